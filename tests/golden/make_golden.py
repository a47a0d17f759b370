"""Regenerates tests/golden/oracle_vectors.json from the oracle (oracle/*.hpp).

These are REGRESSION vectors of the restatement, not reference outputs: the reference cannot be built in this
environment (Boost/htslib/sdsl-lite absent).  The only reference-measured values are the SURVEY.md §8(c) known
answers asserted literally in tests/test_oracle.py.
"""
import json, os, random, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O
from conftest import genome_text, make_genome, make_queries

rng = random.Random(2026)
out = {"neighbors": [], "needle": [], "hunt": {}}
qs = ["ACGTACGTAC", "AAAAAAAAAAAA", "ACGTNACGTTGCA", "TCTCTGCACACACGTTGTAC", "GATTACAGATTACA", "CCCCCCCCCCGGGGGGGGGG"]
for q in qs:
    for indel in (False, True):
        for d in (0, 1):
            out["neighbors"].append({"query": q, "distance": d, "indel": indel, "set": O.neighbors(q, d, indel)})
out["neighbors"].append({"query": "ACGTTGCAAC", "distance": 2, "indel": True, "set": O.neighbors("ACGTTGCAAC", 2, True)})
for _ in range(60):
    n = rng.randint(10, 24)
    q = "".join(rng.choice("ACGT") for _ in range(n))
    w = list(q)
    for _e in range(rng.randint(0, 2)):
        k = rng.randrange(len(w)); r = rng.random()
        if r < .33: w[k] = rng.choice("ACGT")
        elif r < .66: del w[k]
        else: w.insert(k, rng.choice("ACGT"))
    w = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 2))) + "".join(w) + "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 2)))
    out["needle"].append({"a1": w, "a2": q, "out": list(O.needle(w, q))})
out["needle"].append({"a1": "GACGTTCGTACA", "a2": "ACGTACGTAC", "out": list(O.needle("GACGTTCGTACA", "ACGTACGTAC"))})
gargs = [77, 3, 12000]
seqs = make_genome(*gargs)
text = genome_text(seqs)
names = ["chrA", "chrB", "chrC"]
O.build_fm9(text, "/tmp/golden.fm9")
ix = O.Index("/tmp/golden.fm9")
cases = []
t = text.decode()
edge = [seqs[0][:20], seqs[0][-20:], seqs[1][:19], seqs[2][-21:], seqs[1][1:21], "acgtnacgtacgtacgtacg", "ACGTAC", seqs[0][100:118], "A" * 20, "ACGU" * 5]
for params, queries in [
    ({"distance": 1}, make_queries(1, text, 12) + edge),
    ({"distance": 0}, make_queries(2, text, 6, lens=(18,)) + edge[:4]),
    ({"distance": 1, "hamming": True}, make_queries(3, text, 8) + edge),
    ({"distance": 1, "forward_only": True, "max_locations": 2}, make_queries(4, text, 8, lens=(12, 15)) + edge[:5]),
]:
    qn = ["q%03d" % i if i % 3 else "" for i in range(len(queries))]
    js, _ = ix.hunt([len(s) + 1 for s in seqs], names, queries, qnames=qn, genome="genome.fa.gz", **params)
    cases.append({"params": params, "queries": queries, "qnames": qn, "json": js})
out["hunt"] = {"genome_args": gargs, "names": names, "cases": cases}
json.dump(out, open(os.path.join(HERE, "oracle_vectors.json"), "w"), indent=0)
print("written", sum(len(c["queries"]) for c in cases), "hunt queries")
