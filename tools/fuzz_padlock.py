#!/usr/bin/env python3
"""Randomised differential run of `dicey padlock` (the repo's binary) against the oracle (restated padlock.h/gtf.h + the
reference's own thal.h) on the scenario of tests/padlock_fixture.py with random option combinations."""
import gzip, os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O, padlock_fixture as F

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = random.Random(seed)
DICEY = os.environ.get("DICEY_BIN", os.path.join(ROOT, "dicey_amd", "dicey"))  # DICEY_BIN: e.g. a tools/hostemu-linked binary
d = tempfile.mkdtemp(prefix="fuzz_padlock_")
sc = F.build(d)
if "DICEY_BIN" in os.environ: O.build_fm9(sc["text"], sc["fm9"])  # the emulator build has no index builder
else: assert subprocess.run([DICEY, "index", sc["fa"]], capture_output=True).returncode == 0
orc = O.Index(sc["fm9"])
bad = 0
for c in range(nconf):
    args, kw = [], {}
    arm = rng.choice([15, 18, 20, 20, 22, 24, 25, 28]); args += ["-m", str(arm)]; kw["armlen"] = arm
    ham = rng.random() < 0.5
    if ham: args += ["-n"]; kw["hamming"] = True
    dist = rng.choice([0, 1, 1, 2])
    if dist == 2 and not ham: dist = 1  # the oracle's edit-distance-2 neighbourhoods cost seconds per arm on the CPU
    args += ["-d", str(dist)]; kw["distance"] = dist
    if rng.random() < 0.4: args += ["-p"]; kw["probe_mode"] = True
    if rng.random() < 0.5: args += ["-v"]; kw["overlapping"] = True
    lo, hi = rng.choice([(0.4, 0.6), (0.3, 0.7), (0.2, 0.8), (0.45, 0.55)]); args += ["--gcmin", str(lo), "--gcmax", str(hi)]; kw["gcmin"] = lo; kw["gcmax"] = hi
    z = rng.choice([1, 2, 5, 10]); args += ["-z", str(z)]; kw["tmdiff"] = z
    if rng.random() < 0.3:
        mv, dv = rng.choice([(40.0, 2.5), (100.0, 0.0), (50.0, 3.0)]); args += ["--monovalent", str(mv), "--divalent", str(dv)]; kw["mv"] = mv; kw["dv"] = dv
    inp = rng.choice(["@genes.lst", "all", "ENSG03", "@custom.fa"])
    if inp == "@genes.lst": kw["genes"] = F.FOUR
    elif inp == "all": kw["compute_all"] = True
    elif inp == "ENSG03": kw["genes"] = ["ENSG03"]
    else:
        kw["input_fasta"] = True
        if rng.random() < 0.5: args += ["-e"]; kw["absent"] = True
    case = ("fuzz%d" % c, args, inp, kw)
    out, js = os.path.join(d, "out.tsv"), os.path.join(d, "out.json.gz")
    for x in (out, js):
        if os.path.exists(x): os.remove(x)
    infile = os.path.join(d, inp[1:]) if inp.startswith("@") else inp
    r = subprocess.run([DICEY, "padlock", "-g", sc["fa"], "-t", sc["gtf"], "-b", sc["bar"], "-i", O.PRIMER3_CONFIG, "-o", out, "-j", js, *args, infile],
                       capture_output=True, text=True, timeout=900)
    tsv = open(out).read() if os.path.exists(out) else ""
    jt = gzip.open(js, "rt").read() if os.path.exists(js) else ""
    wt, wj, we, wrc = F.oracle_run(orc, sc, case, out, js)
    ok = (r.returncode, tsv, jt) == (wrc, wt, wj)
    print("conf", c, args, inp, "rows", max(0, wt.count("\n") - 1), "ok" if ok else "MISMATCH rc %d/%d tsv %s json %s | %s" % (r.returncode, wrc, tsv == wt, jt == wj, r.stderr[-300:]))
    bad += not ok
print("failing configurations:", bad)
