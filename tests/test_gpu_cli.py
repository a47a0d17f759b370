"""Process seam: the `dicey` binary of this repo — same argv surface, same JSON — against the
oracle's restatement of hunter.h:99-160,291-444, byte for byte."""
import gzip
import os
import subprocess

import pytest

import oracle_lib as O
from conftest import genome_text, make_genome, make_queries

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DICEY = os.path.join(ROOT, "dicey_amd", "dicey")


@pytest.fixture(scope="module")
def cli_genome(tmp_path_factory):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dicey_amd", "cli"), "-s"])
    d = tmp_path_factory.mktemp("cli")
    seqs = make_genome(55, 3, 20000)
    names = ["chr1", "chr2 some description", "scaffold_3"]
    fa = d / "genome.fa.gz"
    with gzip.open(fa, "wt") as f:
        for n, s in zip(names, seqs):
            f.write(">" + n + "\n")
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60].lower() if i % 120 else s[i:i + 60])
                f.write("\n")
    r = subprocess.run([DICEY, "index", str(fa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fm9 = str(d / "genome.fa.fm9")  # genome.parent_path()/stem + ".fm9" (hunter.h:254-255)
    assert os.path.exists(fm9)
    # the GPU-built file must equal the oracle's construction on the text index.h:97-115 defines
    ref = str(d / "oracle.fm9")
    O.build_fm9(genome_text(seqs), ref)
    assert open(fm9, "rb").read() == open(ref, "rb").read()
    return {"fa": str(fa), "fm9": fm9, "seqs": seqs, "names": ["chr1", "chr2", "scaffold_3"], "dir": d,
            "seqlen": [len(s) + 1 for s in seqs], "text": genome_text(seqs)}


def _oracle_json(g, queries, qnames, **kw):
    ix = O.Index(g["fm9"])
    js, _ = ix.hunt(g["seqlen"], g["names"], queries, qnames=qnames, genome=g["fa"], **kw)
    return js


def test_hunt_literal_sequence_json_identical(cli_genome):
    g = cli_genome
    q = g["seqs"][1][500:520]
    r = subprocess.run([DICEY, "hunt", "-g", g["fa"], q], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == _oracle_json(g, [q], [""], distance=1)
    # option forms program_options accepts: -d0, --distance=0, long-option prefix, switches
    for extra, kw in [(["-d0"], dict(distance=0)), (["--distance=0", "-n"], dict(distance=0, hamming=True)),
                      (["--dist", "1", "--forward", "-m", "2"], dict(distance=1, forward_only=True, max_locations=2))]:
        r = subprocess.run([DICEY, "hunt", *extra, "-g", g["fa"], q], capture_output=True, text=True)
        assert r.stdout == _oracle_json(g, [q], [""], **kw), extra


def test_hunt_fasta_batch_and_gz_outfile(cli_genome):
    g = cli_genome
    qs = make_queries(8, g["text"], 200) + ["ACGTAC", "acgtnnacgtacgtacgtac", g["seqs"][0][:20], g["seqs"][2][-20:]]
    names = ["q%d extra words" % i for i in range(len(qs))]
    fa = g["dir"] / "queries.fa"
    with open(fa, "w") as f:
        for n, s in zip(names, qs):
            f.write(">%s\n%s\n\n%s\n" % (n, s[:7], s[7:]))
    want = _oracle_json(g, qs, names, distance=1)
    r = subprocess.run([DICEY, "hunt", "-g", g["fa"], str(fa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want
    out = g["dir"] / "hits.json.gz"
    r = subprocess.run([DICEY, "hunt", "-o", str(out), "-g", g["fa"], str(fa)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == ""
    ix = O.Index(g["fm9"])
    want_o, _ = ix.hunt(g["seqlen"], g["names"], qs, qnames=names, genome=g["fa"], outfile=str(out), distance=1)
    assert gzip.open(out, "rt").read() == want_o  # one gzip member per query, concatenated (hunter.h:169)


def test_error_paths_match_reference_messages(cli_genome):
    g = cli_genome
    r = subprocess.run([DICEY, "hunt", "-g", "/no/such/genome.fa.gz", "ACGTACGTACGT"], capture_output=True, text=True)
    assert r.returncode == 1
    assert r.stdout == '{"errors": [{"title":"Error: Genome does not exist!","type":"error"}]}\n'
    r = subprocess.run([DICEY, "hunt", "ACGTACGTACGT"], capture_output=True, text=True)
    assert r.returncode == 255 and r.stdout.startswith("Usage: dicey hunt [OPTIONS] -g Danio_rerio.fa.gz CATTACTAACATCAGT")
    notfa = g["dir"] / "notfasta.txt"
    notfa.write_text("hello\n")
    r = subprocess.run([DICEY, "hunt", "-g", g["fa"], str(notfa)], capture_output=True, text=True)
    assert r.returncode == 1 and "Error: Input file is not in FASTA format!" in r.stdout


def test_json_strings_with_special_characters(tmp_path):
    """chromosome and query names with quotes, backslashes, control characters and UTF-8: the binary's writer against the
    oracle, whose objects are dumped by the reference's own nlohmann::json (oracle/_ref/libjsonref.so) when present"""
    seqs = make_genome(91, 3, 8000)
    names = ['chr"1"', "back\\slash|x", "ünï©ødé_漢字😀"]
    fa = tmp_path / 'g "x".fa.gz'
    with gzip.open(fa, "wt") as f:
        for n, s in zip(names, seqs):
            f.write(">" + n + " description\n" + s + "\n")
    assert subprocess.run([DICEY, "index", str(fa)], capture_output=True).returncode == 0
    fm9 = str(fa)[:-3][:-3] + ".fm9" if False else str(tmp_path / 'g "x".fa.fm9')
    assert os.path.exists(fm9)
    qs = [seqs[0][100:120], seqs[1][300:322], seqs[2][50:62], "ACGTAC"]
    qn = ['q "one"\twith tab', "two\\three \x01\x1f ctl", "ünï 漢字 😀", "short\x7f"]
    qf = tmp_path / "q.fa"
    with open(qf, "w") as f:
        for n, s in zip(qn, qs):
            f.write(">%s\n%s\n" % (n, s))
    g = {"fm9": fm9, "fa": str(fa), "seqlen": [len(s) + 1 for s in seqs], "names": names}
    for extra, kw in [([], dict(distance=1)), (["-d", "2", "-x", "40"], dict(distance=2, max_neighborhood=40))]:
        r = subprocess.run([DICEY, "hunt", *extra, "-g", str(fa), str(qf)], capture_output=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout.decode() == _oracle_json(g, qs, qn, **kw)
    if os.path.exists(O.REF_JSON):
        assert O.ref_json_in_use()


def test_baseline_config0_single_18mer_hamming0_on_a_4_6_mb_genome(tmp_path):
    """BASELINE.json configs[0] / SURVEY.md §8(d) C1: `dicey hunt -n -d 0 -g <genome> <18-mer>` on one 4.64 Mb sequence (the
    E. coli K-12 size class; no real genome is available offline, so i.i.d. ACGT with a planted 5 kb duplication),
    18-mer = T[1000000..1000018).  Index by `dicey index`, JSON byte-identical to the oracle, coordinates checked directly."""
    import json
    import random
    rng = random.Random(4641652)
    n = 4641652
    seq = bytearray(rng.choices(b"ACGT", k=n))
    seq[3000000:3005000] = seq[999000:1004000]  # the 18-mer occurs twice
    seq = seq.decode()
    fa = tmp_path / "ecoli_like.fa.gz"
    with gzip.open(fa, "wt", compresslevel=1) as f:
        f.write(">U00096.3 synthetic\n")
        for i in range(0, n, 80):
            f.write(seq[i:i + 80])
            f.write("\n")
    r = subprocess.run([DICEY, "index", str(fa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    q = seq[1000000:1000018]
    r = subprocess.run([DICEY, "hunt", "-n", "-d", "0", "-g", str(fa), q], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    g = {"fa": str(fa), "fm9": str(tmp_path / "ecoli_like.fa.fm9"), "seqlen": [n + 1], "names": ["U00096.3"]}
    assert r.stdout == _oracle_json(g, [q], [""], distance=0, hamming=True)
    doc = json.loads(r.stdout)
    starts = sorted((h["chr"], h["start"], h["strand"]) for h in doc["data"])
    want = [("U00096.3", 1000001, "+"), ("U00096.3", 3001001, "+")]
    assert [s for s in starts if s[2] == "+"] == want, starts


def test_large_input_runs_chunks_in_a_pipeline(cli_genome):
    """More than 2^18 queries: the binary cuts the input into chunks and keeps two batches in flight (dg_hunt_submit / dg_hunt_wait
    on two handles of one resident index).  Output must be the bytes of the chunk-after-chunk form, in query order; a sample of the
    lines is held against the checker."""
    g = cli_genome
    base = make_queries(9, g["text"], 3000)
    qs = (base * 90)[:270000]  # > 2^18 -> three chunks of 2^17
    fa = g["dir"] / "many.fa"
    with open(fa, "w") as f:
        for i, s in enumerate(qs):
            f.write(">m%d\n%s\n" % (i, s))
    env = dict(os.environ, DICEY_KMER_K="9")  # a large input asks for the K-mer table: keep it tiny on this 60 kb genome
    r = subprocess.run([DICEY, "hunt", "-g", g["fa"], str(fa)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    r2 = subprocess.run([DICEY, "hunt", "-g", g["fa"], str(fa)], capture_output=True, text=True, env=dict(env, DICEY_NO_PIPELINE="1"))
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert r.stdout == r2.stdout
    lines = r.stdout.split("\n")
    assert len(lines) == len(qs) + 1 and lines[-1] == ""
    pick = list(range(0, 50)) + list(range(131050, 131100)) + list(range(269950, 270000))
    want = _oracle_json(g, [qs[i] for i in pick], ["m%d" % i for i in pick], distance=1).split("\n")
    assert [lines[i] for i in pick] == want[:-1]


def test_chunks_refused_for_their_size_are_answered_in_halves_and_none_is_dropped(cli_genome):
    """r03 advice: a chunk the library refuses with DG_ELIMIT (its capped neighbourhoods would not fit the memory budget) is answered
    in halves by the blocking call — and the chunks BEHIND it must still be submitted.  Three chunks of cap-prone queries (-x 50: every
    20-mer's 145-string neighbourhood reaches the cap) under a 1 MB budget: every chunk is refused, every query must be answered."""
    g = cli_genome
    base = make_queries(19, g["text"], 2000, (20,))
    qs = (base * 140)[:270000]
    fa = g["dir"] / "capped_many.fa"
    with open(fa, "w") as f:
        for i, s in enumerate(qs):
            f.write(">c%d\n%s\n" % (i, s))
    env = dict(os.environ, DICEY_KMER_K="9", DICEY_CAP_BUDGET_MB="1")
    r = subprocess.run([DICEY, "hunt", "-x", "50", "-g", g["fa"], str(fa)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.split("\n")
    assert len(lines) == len(qs) + 1 and lines[-1] == ""
    pick = list(range(0, 30)) + list(range(131060, 131090)) + list(range(269970, 270000))
    want = _oracle_json(g, [qs[i] for i in pick], ["c%d" % i for i in pick], distance=1, max_neighborhood=50).split("\n")
    assert [lines[i] for i in pick] == want[:-1]


def test_index_verify_accepts_its_file_and_refuses_foreign_ones(cli_genome, tmp_path):
    """`dicey index --verify genome.fa.gz` (r06, A13): the acceptance procedure for an index file — sections on the host
    (dg_fm9_check), the device's own checks at open, the whole text against the FASTA, sampled count / locate.  The file this build
    wrote passes; files that are well-formed but say something else (another child order in the Huffman tree, a rank word of another
    shape, a shifted C[]) are refused BY NAME and never load (dg_index_open fails on each); an index of another genome is refused at
    the first differing text position."""
    import shutil
    import struct
    import dicey_amd
    g = cli_genome
    r = subprocess.run([DICEY, "index", "--verify", g["fa"]], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("Verified: ") and "[4/4]" in r.stderr
    rep = dicey_amd.check_fm9(g["fm9"])
    secs = {s["name"]: (s["offset"], s["bytes"]) for s in rep["sections"]}
    data = open(g["fm9"], "rb").read()
    fa2 = str(tmp_path / "genome.fa.gz")
    shutil.copy(g["fa"], fa2)
    fm2 = str(tmp_path / "genome.fa.fm9")

    def swap_children(b):
        o = secs["wt byte_tree nodes"][0] + 8
        b[o + 18:o + 20], b[o + 20:o + 22] = b[o + 20:o + 22], b[o + 18:o + 20]

    def rank_word(b):
        o = secs["wt rank_support_v"][0] + 8 + 16
        struct.pack_into("<Q", b, o, struct.unpack_from("<Q", b, o)[0] + 1)

    def c_shift(b):
        o = secs["alphabet C"][0] + 8 + 16
        struct.pack_into("<Q", b, o, struct.unpack_from("<Q", b, o)[0] + 1)

    def bv_bit(b):
        b[secs["wt bit_vector"][0] + 8 + 100] ^= 4

    def sa_sample(b):   # a sample in the middle of the vector: in range, wrong
        o = secs["sa_samples"][0] + 9 + secs["sa_samples"][1] // 2
        b[o] ^= 1

    for name, fn, word in (("children", swap_children, "byte_tree"), ("rank", rank_word, "rank_support_v"), ("C", c_shift, "alphabet C"),
                           ("bit", bv_bit, "rank_support_v"), ("sample", sa_sample, None)):
        b = bytearray(data)
        fn(b)
        open(fm2, "wb").write(bytes(b))
        with pytest.raises(Exception):          # none of them loads
            dicey_amd.FmIndex(fm2).close()
        r = subprocess.run([DICEY, "index", "--verify", fa2], capture_output=True, text=True)
        assert r.returncode == 1 and "REFUSED" in r.stderr, (name, r.stderr)
        if word:
            assert word in r.stderr, (name, r.stderr)
    # the index of another genome: well-formed, self-consistent, refused against this FASTA
    other = make_genome(56, 3, 20000)
    O.build_fm9(genome_text(other), fm2)
    r = subprocess.run([DICEY, "index", "--verify", fa2], capture_output=True, text=True)
    assert r.returncode == 1 and "text position" in r.stderr, r.stderr
    # ... and a hunt against a perturbed file reports the reference's error line, it does not answer
    b = bytearray(data)
    swap_children(b)
    open(fm2, "wb").write(bytes(b))
    r = subprocess.run([DICEY, "hunt", "-g", fa2, g["seqs"][0][100:120]], capture_output=True, text=True)
    assert r.returncode == 1 and "FM-Index cannot be loaded" in r.stdout


def test_output_to_a_regular_file_equals_the_piped_output(cli_genome):
    """r06: stdout goes through a writer thread (the next chunk is formatted while the previous one is written); a pipe and a regular
    file — also one that already holds bytes — must receive the same bytes, in query order; multi-line FASTA records in the input
    (sequences joined, empty lines skipped: hunter.h:272-283).  DICEY_RESIDENT_LAYOUTS opens the index with the layouts a one-shot
    process leaves out (DG_OPEN_COMPACT): same output."""
    g = cli_genome
    base = make_queries(29, g["text"], 3000, (18, 20, 23))
    qs = (base * 100)[:280000]
    fa = g["dir"] / "many2.fa"
    with open(fa, "w") as f:
        for i, s in enumerate(qs):
            if i % 7 == 0:   # a record whose sequence spans two lines, and an empty line
                f.write(">m%d\n%s\n\n%s\n" % (i, s[:9], s[9:]))
            else:
                f.write(">m%d\n%s\n" % (i, s))
    env = {k: v for k, v in os.environ.items() if not k.startswith("DICEY_")}
    piped = subprocess.run([DICEY, "hunt", "-g", g["fa"], str(fa)], capture_output=True, env=env)
    assert piped.returncode == 0, piped.stderr[-1000:]
    assert piped.stdout.count(b"\n") == len(qs)
    out = g["dir"] / "redirected.jsonl"
    with open(out, "wb") as o:
        o.write(b"x" * 1234)
        o.flush()
        r = subprocess.run([DICEY, "hunt", "-g", g["fa"], str(fa)], stdout=o, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr[-1000:]
    data = open(out, "rb").read()
    assert data[:1234] == b"x" * 1234 and data[1234:] == piped.stdout
    with open(out, "wb") as o:
        r = subprocess.run([DICEY, "hunt", "-g", g["fa"], str(fa)], stdout=o, stderr=subprocess.PIPE, env=dict(env, DICEY_RESIDENT_LAYOUTS="1"))
    assert r.returncode == 0 and open(out, "rb").read() == piped.stdout
    want = _oracle_json(g, [qs[i] for i in range(0, 21)], ["m%d" % i for i in range(0, 21)], distance=1).encode().split(b"\n")
    assert piped.stdout.split(b"\n")[:21] == want[:21]
