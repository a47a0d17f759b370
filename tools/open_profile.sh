#!/bin/bash
# GPU box: kernel trace of one `dicey hunt` process (index open + derivation + one batch) on the bench genome
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/open
rm -rf $OUT; mkdir -p $OUT
timeout 900 python bench.py --keep-index --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
python - "$FM9" <<'PY'
import json, sys, glob, os
fm9 = sys.argv[1]
meta = json.load(open(glob.glob(fm9 + ".hunt_d1.*.meta.json")[0]))
base = "/dev/shm/cli_genome.fa"
with open(base + ".gz.fai", "w") as f:
    off = 0
    for i, l in enumerate(meta["lens"]):
        f.write("s%d\t%d\t%d\t60\t61\n" % (i, l, off)); off += l + l // 60 + 10
open(base + ".gz", "wb").write(b"\x1f\x8b placeholder")
if not os.path.lexists(base + ".fm9"): os.symlink(fm9, base + ".fm9")
with open("/dev/shm/cli_queries.fa", "w") as f:
    for i, q in enumerate(meta["queries"][0]):
        f.write(">q%06d\n%s\n" % (i, q))
PY
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace --output-format csv -- env DICEY_TIMING=1 $GRAFT_REPO_ROOT/dicey_amd/dicey hunt -g /dev/shm/cli_genome.fa.gz /dev/shm/cli_queries.fa > /dev/shm/cli_out.jsonl 2> $GRAFT_REPO_ROOT/$OUT/time.txt)
python - <<'PY'
import csv, re
rows = list(csv.DictReader(open("gpurun_out/open/trace/trace_kernel_stats.csv")))
tot = 0
for r in rows:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    n = re.sub(r"<.*", "<>", n) if "rocprim" in n else n
    t = float(r["TotalDurationNs"]) / 1e6
    tot += t
    if t > 1: print("%-60s calls %4s total %9.2f ms" % (n[:60], r["Calls"], t))
print("all kernels %.1f ms" % tot)
PY
grep "dicey timing" $OUT/time.txt
rm -f /dev/shm/dicey_bench_* /dev/shm/cli_*
