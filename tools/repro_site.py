import os, sys, random, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import dicey_amd, p3config
rng = random.Random(5)
g = [rng.choice("ACGT") for _ in range(200000)]
sites = ["GACGCATGTCACATGGCATAATAAT", "GACGCATGTCACTGGTCATCACCTT", "GACGCATGTCAGTGGTCTATCAAGG", "GACGCATGTCGATGGCTATGAATCC", "GACGCGTGTCAATGGTCCCTTGCCA",
         "GACGCATGTCAATGGCACTAATA"]
for k, s in enumerate(sites):
    p = 10000 + 20000 * k
    g[p:p + len(s)] = list(s)
text = "".join(g)
fm9 = "/tmp/repro.fm9"
dicey_amd.build_index((text + "\n").encode(), fm9)
ix = dicey_amd.FmIndex(fm9)
th = dicey_amd.Thal(p3config.config_dir())
prim = ["TATTAGTGCCATTGACATGCGTC", "AATGTGTCCTTTGTAACCAATTA"]
s, mt, fl, nh = dicey_amd.search_sites(ix, th, prim, [len(text) + 1], cut_temp=30.0)
for x in s: print(json.dumps({k: (v.hex() if isinstance(v, float) else v) for k, v in x.items()}, sort_keys=True))
print("nh", nh)
