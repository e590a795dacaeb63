#ifndef demo_ops_H
#define demo_ops_H

/* An operator group as a maintainer of the reference would write one: the five functions per
 * operator behind dspprototypes(), and the rows for the driver's table.  Test asset of
 * tests/test_plugin_boundary.py (compiled with the driver: make EXTRA_OPS_HEADER=... EXTRA_OPS_SRCS=...). */

dspprototypes(op_demo_lift)
dspprototypes(op_demo_snapshot)

#define GDSP_EXTRA_DSPTABLE_ROWS \
	dspinforecord("demolift"     , op_demo_lift)     , dspinfoalias ("demo_lift") , \
	dspinforecord("demosnapshot" , op_demo_snapshot)

#endif
