"""Rewrite tests/dropin/mpi_reftests_expected.json from the logs of a run of Ginkgo's own MPI test
binaries on the GPU (tools/run_mpi_reftests.sh: <dir>/<suite>_mpi_hip.log is rank 0's report,
<dir>/<suite>.rank<k>.log the other ranks').  A test counts as failing when it fails on ANY rank.
  python tools/update_mpi_reftests_expected.py gpurun_out/r03s17/mpi"""
import glob
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "tests", "dropin", "mpi_reftests_expected.json")
exp = json.load(open(path)) if os.path.exists(path) else {}
src = sys.argv[1]
RANKS = {"distributed_row_gatherer": 6}   # test/mpi/distributed/CMakeLists.txt: MPI_SIZE 6; 3 elsewhere
for f in sorted(glob.glob(os.path.join(src, "*_mpi_hip.log"))):
    suite = os.path.basename(f)[:-len("_mpi_hip.log")]
    txt0 = open(f, errors="replace").read()
    ran = re.search(r"^\[==========\] (\d+) tests ran", txt0, re.M)
    if not ran:
        print(f"{suite}: did not run to its end, left alone")
        continue
    reasons = {}
    for g in [f] + sorted(glob.glob(os.path.join(src, suite + ".rank*.log"))):
        txt = open(g, errors="replace").read()
        failed = set(re.findall(r"^\[  FAILED  \] (.+)$", txt, re.M))
        for t in sorted(failed):
            if re.match(r"\d+ tests?, listed below:", t) or t.startswith("on a rank other than 0"):
                continue
            m = re.search(r"^\[ RUN      \] " + re.escape(t) + r"\n(.*?)^\[  FAILED  \] " + re.escape(t), txt,
                          re.M | re.S)
            why = "assertion"
            if m:
                r = re.search(r"feature (\S+) is part of the hip module", m.group(1))
                e = re.search(r"C\+\+ exception with description \"([^\"]*)\"", m.group(1))
                why = f"NotCompiled: {r.group(1)}" if r else e.group(1)[-160:] if e else "assertion"
            reasons.setdefault(t, why)
    print(f"{suite}: ran {ran.group(1)}, failing {len(reasons)}")
    exp[suite] = {"ranks": RANKS.get(suite, 3), "ran": int(ran.group(1)), "known_failures": reasons}
json.dump(exp, open(path, "w"), indent=1, sort_keys=True)
total = sum(v["ran"] for v in exp.values())
bad = sum(len(v["known_failures"]) for v in exp.values())
print(f"{total - bad} of {total} tests pass")
