/* check_output_format.c -- the driver's hand-rolled "%.*f" (put_fixed, genodsp_amd/host/genodsp_hip.c) against
 * snprintf on 20 M values.  tools/check_output_format.sh extracts the function from the driver and builds this. */
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
typedef double valtype;
#define valtypeFmtPrec "%.*f"
#include "put_fixed.inc"
/* put_fixed returns NULL for what the driver prints through printf itself (huge values, precisions beyond 9, NaN) */
static void fmt (char* a, double v, int p) { char* q = put_fixed (a, v, p);  if (q == NULL) snprintf (a, 512, "%.*f", p, v);  else *q = 0; }
int main(int argc, char** argv){ long iters = (argc > 1)? atol (argv[1]) : 20000000; double vals[]={0,-0.0,1,-1,12,1e14,999999999999999.0,1e15,-1e15,0.5,2.5,-3.75,1e300,123456789012.0,NAN,INFINITY,-7,4294967296.0,9007199254740992.0,0.125,0.375,1e-5,-1e-5,1e-300,5e-324,0.0005,0.0015,0.0025,2.675,1.005,7.9999e12,-7.9999e12,8e12,0.045,1.45,2.5e-7,0.15,0.25,0.35};
 long bad=0; for(int p=0;p<=11;p++) for(unsigned i=0;i<sizeof(vals)/8;i++){ char a[512],b[512]; fmt(a,vals[i],p); snprintf(b,512,"%.*f",p,vals[i]); if(strcmp(a,b)){bad++; printf("MISMATCH p=%d %s %s\n",p,a,b);} }
 srand(1);
 for(long k=0;k<iters;k++){ double v; int kind=rand()%6;
   if(kind==0) v=(rand()%2000001-1000000)*0.25;
   else if(kind==1) v=(rand()/(double)RAND_MAX)*100;
   else if(kind==2) v=((rand()%200001)-100000)/1000.0;     /* decimal-looking: many near-ties */
   else if(kind==3) v=((rand()%2001)-1000)/16.0;
   else if(kind==4) v=ldexp((double)rand(), -(rand()%80));
   else { union{double d; uint64_t u;} c; c.u=((uint64_t)rand()<<33)^((uint64_t)rand()<<11)^rand(); v=c.d; }
   int p=rand()%10; char a[1024],b[1024]; fmt(a,v,p); snprintf(b,1024,"%.*f",p,v); if(strcmp(a,b)){bad++; if(bad<8)printf("MISMATCH p=%d %.17g : %s %s\n",p,v,a,b);} }
 printf("bad=%ld\n",bad); return 0; }
