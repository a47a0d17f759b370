import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build what is missing (a checkout without the git-ignored artefacts): the gfx950 library, the host binary, the
    oracle.  hipcc cross-compiles without a GPU, so this works in the build container and on the GPU box alike."""
    import subprocess
    need = [(os.path.join(ROOT, "dicey_amd", "libdiceygpu.so"), ["make", "-C", os.path.join(ROOT, "dicey_amd", "csrc"), "-s", "-j4"]),
            (os.path.join(ROOT, "dicey_amd", "libdiceygather.so"), ["make", "-C", os.path.join(ROOT, "dicey_amd", "csrc"), "-s", "-j4"]),
            (os.path.join(ROOT, "dicey_amd", "dicey"), ["make", "-C", os.path.join(ROOT, "dicey_amd", "cli"), "-s"]),
            (os.path.join(ROOT, "oracle", "liboracle.so"), ["make", "-C", os.path.join(ROOT, "oracle"), "-s"])]
    for artefact, cmd in need:
        if not os.path.exists(artefact):
            subprocess.call(cmd)


# The development build that reads the test switches (dicey_amd/csrc/experiments.hpp): the product library ignores them, so every test
# that forces a code path through an environment variable opens its index on this library (open_index / exp_lib).
EXP_LIB = os.path.join(ROOT, "dicey_amd", "variants", "libdiceygpu_exp.so")
EXP_VARS = ("DICEY_NO_BAND_VERIFY", "DICEY_CAP_HOST", "DICEY_NO_FUSED_SELECT", "DICEY_NO_FUSED_SELECT2", "DICEY_NO_PREP_FUSION", "DICEY_NO_PRE5_D2",
            "DICEY_NO_FLAT_HAMMING2", "DICEY_NO_N_WINDOW", "DICEY_NO_LONG2", "DICEY_DEBUG_CAPS", "DICEY_FUSED_LCAP", "DICEY_VERIFY_CH", "DICEY_EXP",
            "DICEY_NO_KMER_FILTER", "DICEY_NO_NRUN_PRUNE", "DICEY_NO_DIRECT_CTX", "DICEY_NO_PRE5", "DICEY_NO_SAX", "DICEY_NO_PLV", "DICEY_NO_SA_MINIMA",
            "DICEY_NO_LDS_TABLES", "DICEY_NO_WAVE_THAL", "DICEY_DEBUG_THAL_REDO")
_exp = {"lib": None}


def _sources_mtime():
    import glob
    files = glob.glob(os.path.join(ROOT, "dicey_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "dicey_amd", "csrc", "*.hpp")) + \
        [os.path.join(ROOT, "include", "dicey_gpu.h")]
    return max(os.path.getmtime(f) for f in files)


def build_exp_lib():
    """tools/build_variant.sh exp -DDG_EXPERIMENTS when the library is missing or older than the kernel sources"""
    import subprocess
    if not os.path.exists(EXP_LIB) or os.path.getmtime(EXP_LIB) < _sources_mtime():
        subprocess.check_call([os.path.join(ROOT, "tools", "build_variant.sh"), "exp", "-DDG_EXPERIMENTS"], stdout=subprocess.DEVNULL)
    return EXP_LIB


def exp_lib():
    if _exp["lib"] is None:
        from dicey_amd import _capi
        _exp["lib"] = _capi.load(build_exp_lib())
    return _exp["lib"]


def open_index(fm9, **kw):
    """dicey_amd.FmIndex on the product library — or, when a test switch is set in the environment, on the development build"""
    import dicey_amd
    return dicey_amd.FmIndex(fm9, _lib=exp_lib() if any(k in os.environ for k in EXP_VARS) else None, **kw)


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need an MI355X: on a box without a HIP device they are skipped (not failed), so a plain `pytest` run is
    green wherever it runs.  (DICEY_LIB — a development build of the library — lifts the skip.)"""
    if os.environ.get("DICEY_LIB") or not any("gpu" in it.keywords for it in items):
        return
    try:
        from dicey_amd import _capi
        have = _capi.load().dg_device_count() > 0
    except Exception:
        have = False
    if not have and not os.path.exists("/dev/kfd"):  # a box with the GPU driver node never skips: a broken run must fail loudly
        skip = pytest.mark.skip(reason="no HIP device on this box (the product has no CPU path)")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


def make_genome(seed, nchr, length, nrate=0.002, repeats=True, iupac=False):
    """Small synthetic multi-chromosome genome with N runs, copied segments and homopolymers."""
    rng = random.Random(seed)
    seqs = []
    for _ in range(nchr):
        s = []
        while len(s) < length:
            r = rng.random()
            if r < nrate:
                s.extend("N" * rng.randint(1, 150))
            elif repeats and r < nrate + 0.001 and len(s) > 200:
                a = rng.randrange(len(s) - 100)
                s.extend(s[a:a + rng.randint(20, 100)])
            elif repeats and r < nrate + 0.0015:
                s.extend(rng.choice("ACGT") * rng.randint(5, 40))
            elif iupac and r < nrate + 0.002:
                s.append(rng.choice("RYKMSW"))
            else:
                s.append(rng.choice("ACGT"))
        seqs.append("".join(s[:length]))
    return seqs


def genome_text(seqs):
    """The text `dicey index` feeds to sdsl::construct (src/index.h:105-113)."""
    return ("\n".join(seqs) + "\n").encode()


def revcomp(s):
    return s.upper().translate(str.maketrans("ACGTN", "TGCAN"))[::-1]


def make_queries(seed, text, n, lens=(20,), p_genome=0.8):
    """SURVEY §8(d) C2 recipe: mostly genome-sampled (half of them with one random edit), some random."""
    rng = random.Random(seed)
    t = text.decode()
    out = []
    while len(out) < n:
        m = rng.choice(lens)
        if rng.random() < p_genome:
            p = rng.randrange(len(t) - m)
            q = t[p:p + m]
            if "\n" in q:
                continue
            if rng.random() < 0.5:
                k = rng.randrange(len(q))
                r = rng.random()
                if r < 1 / 3:
                    q = q[:k] + rng.choice("ACGT") + q[k + 1:]
                elif r < 2 / 3:
                    q = q[:k] + q[k + 1:]
                else:
                    q = q[:k] + rng.choice("ACGT") + q[k:]
            if rng.random() < 0.3:
                q = revcomp(q)
        else:
            q = "".join(rng.choice("ACGT") for _ in range(m))
        out.append(q)
    return out


@pytest.fixture(scope="session")
def small_genome(tmp_path_factory):
    import oracle_lib as O
    seqs = make_genome(101, 3, 30000, iupac=True)
    text = genome_text(seqs)
    path = str(tmp_path_factory.mktemp("fm") / "small.fm9")
    O.build_fm9(text, path)
    return {"seqs": seqs, "text": text, "fm9": path, "seqlen": [len(s) + 1 for s in seqs],
            "names": ["chr%d" % (i + 1) for i in range(len(seqs))]}
