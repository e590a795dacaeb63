import json, sys, hashlib, tempfile, pathlib
sys.path.insert(0, "tests")
from test_cli_hip import run
d = json.load(open("tests/golden/golden.json"))
cases = d["cases"] if isinstance(d, dict) else d
names = sys.argv[2].split(",")
N = int(sys.argv[1])
for c in cases:
    if c.get("name") not in names:
        continue
    bad = 0
    for k in range(N):
        with tempfile.TemporaryDirectory() as t:
            rc, out, err = run(c["args"], c["stdin"], c["chroms_text"], pathlib.Path(t), c.get("files"))
        if rc != 0 or hashlib.sha256(out.encode()).hexdigest() != c["sha256"]:
            bad += 1
            if bad <= 2:
                body = out.splitlines()
                print(c["name"], "run", k, "rc", rc, "lines", len(body), "want", c["lines"], err[-300:])
                # first differing line vs a good run
                good = globals().get("good_" + c["name"])
                if good:
                    for i, (a, b) in enumerate(zip(body, good)):
                        if a != b:
                            print("  first diff at line", i, repr(a), "vs", repr(b)); break
        else:
            globals()["good_" + c["name"]] = out.splitlines()
    print(c["name"], "bad", bad, "of", N)
