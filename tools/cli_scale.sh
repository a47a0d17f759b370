#!/bin/bash
# GPU box: `dicey hunt` end to end (process seam) on the bench genome: 100 000 queries from a FASTA file to JSON lines.
cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --keep-index --steps 2 --warmup 1 --no-cpu-baseline --pipeline 1 > /tmp/b1.log 2>&1
FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
python - "$FM9" <<'PY'
import json, sys
fm9 = sys.argv[1]
meta = json.load(open(fm9 + ".meta.json"))
base = "/dev/shm/cli_genome.fa"
with open(base + ".gz.fai", "w") as f:
    off = 0
    for i, l in enumerate(meta["lens"]):
        f.write("s%d\t%d\t%d\t60\t61\n" % (i, l, off)); off += l + l // 60 + 10
open(base + ".gz", "wb").write(b"\x1f\x8b placeholder: only the .fai and the .fm9 next to it are read by hunt")
import os
os.symlink(fm9, base + ".fm9") if not os.path.exists(base + ".fm9") else None
with open("/dev/shm/cli_queries.fa", "w") as f:
    for i, q in enumerate(meta["queries"][0]):
        f.write(">q%06d\n%s\n" % (i, q))
print(len(meta["queries"][0]), "queries")
PY
python - <<'PY'
import subprocess, time
t = time.time()
with open("/dev/shm/cli_out.jsonl", "wb") as out:
    r = subprocess.run(["dicey_amd/dicey", "hunt", "-g", "/dev/shm/cli_genome.fa.gz", "/dev/shm/cli_queries.fa"], stdout=out, stderr=subprocess.PIPE)
print("exit", r.returncode, "seconds %.2f" % (time.time() - t), r.stderr[-300:].decode())
PY
wc -l /dev/shm/cli_out.jsonl; head -c 600 /dev/shm/cli_out.jsonl; echo
python - <<'PY'
import json
n = h = 0
for line in open("/dev/shm/cli_out.jsonl"):
    d = json.loads(line); n += 1; h += len(d.get("data", []))
print("lines", n, "hits", h)
PY
rm -f /dev/shm/dicey_bench_* /dev/shm/cli_*
