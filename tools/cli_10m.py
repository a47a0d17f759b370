"""GPU box: `dicey hunt` on a 10 M-query FASTA against the bench genome, three runs, every phase the binary reports (DICEY_TIMING).
    python bench.py --keep-index --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --parity-queries 0 --cli-queries 0
    python tools/cli_10m.py /dev/shm/dicey_bench_*.fm9 [n_queries] [runs] [NAME=value ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

fm9 = sys.argv[1]
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
text, lens = bench.synth_genome(int(3.1e9), 24, seed=1, device=dev)
big = np.concatenate([bench.synth_query_batch(text, 100000, 20, seed=5000 + i) for i in range((nq + 99999) // 100000)])[:nq]
del text
torch.cuda.empty_cache()
meta = {"lens": lens}
extra = dict(kv.split("=", 1) for kv in sys.argv[4:])  # NAME=value ... for the binary's environment only (DIST=2: hunt -d 2)
dist = int(extra.pop("DIST", "1"))
for r in range(runs):
    res = bench.cli_end_to_end(fm9, meta, big, dist, extra_env=extra, bytes_per_query=12000 if dist >= 2 else 700)
    ph = res.pop("index_open_phases_ms", {})
    res.pop("note", None)
    print(json.dumps(res))
    print("   ", "  ".join("%s %.0f" % (k, v) for k, v in ph.items()))
    sys.stdout.flush()
