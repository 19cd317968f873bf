"""Rewrite tests/dropin/reftests_expected.json from the logs of a run of Ginkgo's own test
binaries on the GPU (tools/run_reftests.sh writes <dir>/<suite>.log; any *.log / *.txt named after
a suite works).  Only suites with a log that ran to its end are touched.
  python tools/update_reftests_expected.py gpurun_out/reftests"""
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "tests", "dropin", "reftests_expected.json")
exp = json.load(open(path))
src = sys.argv[1]
for f in sorted(os.listdir(src)):
    suite = os.path.splitext(f)[0]
    if suite not in exp:
        continue
    txt = open(os.path.join(src, f), errors="replace").read()
    ran = re.search(r"^\[==========\] (\d+) tests ran", txt, re.M)
    if not ran:
        print(f"{suite}: did not run to its end, left alone")
        continue
    failed = set(re.findall(r"^\[  FAILED  \] (.+)$", txt, re.M))
    failed = sorted(t for t in failed if not re.match(r"\d+ tests?, listed below:", t))
    reasons = {}
    for t in failed:
        m = re.search(r"^\[ RUN      \] " + re.escape(t) + r"\n(.*?)^\[  FAILED  \] " + re.escape(t), txt, re.M | re.S)
        why = ""
        if m:
            r = re.search(r"feature (\S+) is part of the hip module", m.group(1))
            e = re.search(r"C\+\+ exception with description \"([^\"]*)\"", m.group(1))
            why = f"NotCompiled: {r.group(1)}" if r else e.group(1)[-160:] if e else "assertion"
        reasons[t] = why
    old = exp[suite]
    print(f"{suite}: ran {ran.group(1)} (was {old['ran']}), failing {len(failed)} (was {len(old['known_failures'])})")
    exp[suite] = {"ran": int(ran.group(1)), "known_failures": reasons}
json.dump(exp, open(path, "w"), indent=1, sort_keys=True)
total = sum(v["ran"] for v in exp.values())
bad = sum(len(v["known_failures"]) for v in exp.values())
print(f"{total - bad} of {total} tests pass")
