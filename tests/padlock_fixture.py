"""Synthetic `dicey padlock` scenario shared by the CPU (oracle vs golden) and GPU (binary vs oracle/golden) tests:
a three-sequence genome with a duplicated segment, a near-duplicate (mismatches every 37 nt), an N run, two IUPAC letters
inside an exon and soft-masked bases; a GTF with overlapping exons of two transcripts per gene on both strands, a non-coding gene and an unknown
chromosome; three barcodes for four genes."""
import gzip
import os
import random

NAMES = ["chr1", "chr2", "chrM"]
GENES = [("ENSG01", "GENEA", "chr1", "+", [(1500, 1900), (2100, 2500), (2400, 2800)]),
         ("ENSG02", "GENEB", "chr1", "-", [(6900, 7400), (8950, 9100)]),
         ("ENSG03", "GENEC", "chr2", "+", [(4900, 5700), (11950, 12400)]),
         ("ENSG04", "GENED", "chr2", "-", [(20000, 20030), (21000, 21600)]),
         ("ENSG05", "NONCOD", "chr1", "+", [(15000, 15400)])]
FOUR = ["ENSG01", "ENSG02", "ENSG03", "ENSG04"]
# (label, command-line arguments, positional input, oracle keyword arguments)
CASES = [
    ("genelist", [], "@genes.lst", dict(genes=FOUR)),
    ("hamming_overlapping", ["-n", "-v"], "@genes.lst", dict(genes=FOUR, hamming=True, overlapping=True)),
    ("probe_mode_d2", ["-p", "-d", "2", "-n", "--gcmin", "0.3", "--gcmax", "0.7", "-z", "5"], "ENSG03",
     dict(genes=["ENSG03"], probe_mode=True, distance=2, hamming=True, gcmin=0.3, gcmax=0.7, tmdiff=5)),
    ("all_d0_arm18", ["-d", "0", "-m", "18", "-l", "", "-a", ""], "all", dict(compute_all=True, distance=0, armlen=18, spacerleft="", anchor="")),
    ("noncoding_error", [], "ENSG05", dict(genes=["ENSG05"])),
    ("transcript", ["-u", "transcript_id"], "ENSG02T1", dict(genes=["ENSG02T1"], idname="transcript_id")),
    ("fasta_input", [], "@custom.fa", dict(input_fasta=True)),
    ("fasta_input_absent", ["-e"], "@custom.fa", dict(input_fasta=True, absent=True)),
    ("arm26_general_thal_path", ["-m", "26", "-d", "1", "-n", "-z", "6", "--gcmin", "0.3", "--gcmax", "0.7"], "ENSG03",
     dict(genes=["ENSG03"], armlen=26, hamming=True, tmdiff=6, gcmin=0.3, gcmax=0.7)),
    ("arm31_thal_refuses", ["-m", "31", "-n", "--gcmin", "0.2", "--gcmax", "0.8", "-z", "20"], "ENSG03",
     dict(genes=["ENSG03"], armlen=31, hamming=True, gcmin=0.2, gcmax=0.8, tmdiff=20)),
    ("salt", ["--monovalent", "40", "--divalent", "2.5", "--dna", "100", "--dntp", "0.8"], "ENSG01",
     dict(genes=["ENSG01"], mv=40.0, dv=2.5, dna_conc=100.0, dntp=0.8)),
]


def build(d):
    """Writes the scenario into directory d; returns a dict with paths and the sequences as stored in the FASTA."""
    rng = random.Random(1)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    seqs = [rnd(30000), rnd(24000), rnd(5000)]
    seqs[1] = seqs[1][:5000] + seqs[0][2000:2600] + seqs[1][5600:]          # exact duplicate: arms there are not unique
    nd = list(seqs[0][7000:7300])
    for k in range(10, 300, 37):
        nd[k] = rng.choice("ACGT")
    seqs[1] = seqs[1][:12000] + "".join(nd) + seqs[1][12300:]              # near-duplicate: neighbourhood hits
    seqs[0] = seqs[0][:9000] + "NNNNNNNNNN" + seqs[0][9010:]
    seqs[1] = seqs[1][:5300] + "R" + seqs[1][5301:5420] + "Y" + seqs[1][5421:]   # IUPAC letters inside an exon of ENSG03
    seqs = ["".join(c.lower() if rng.random() < 0.05 else c for c in s) for s in seqs]  # soft-masked bases
    fa = os.path.join(d, "GRCh38_toy.fa.gz")
    with gzip.open(fa, "wt") as f:
        for n, s in zip(NAMES, seqs):
            f.write(">%s some description\n" % n)
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
    gtf = os.path.join(d, "toy.gtf.gz")
    with gzip.open(gtf, "wt") as f:
        f.write("#!genome-build toy\n")
        for gid, sym, chrom, strand, exons in GENES:
            bt = "lncRNA" if gid == "ENSG05" else "protein_coding"
            f.write('%s\ttoy\tgene\t%d\t%d\t.\t%s\t.\tgene_id "%s"; gene_version "1"; gene_name "%s"; gene_source "toy"; gene_biotype "%s";\n'
                    % (chrom, exons[0][0], exons[-1][1], strand, gid, sym, bt))
            for ti in range(2):
                for ei, (s, e) in enumerate(exons):
                    if ti == 1 and ei == 0:
                        s += 20
                    f.write('%s\ttoy\texon\t%d\t%d\t.\t%s\t.\tgene_id "%s"; gene_version "1"; transcript_id "%sT%d"; exon_number "%d"; '
                            'gene_name "%s"; gene_biotype "%s"; transcript_biotype "%s";\n' % (chrom, s, e, strand, gid, gid, ti, ei + 1, sym, bt, bt))
        f.write('chrUn\ttoy\texon\t1\t100\t.\t+\t.\tgene_id "ENSGXX"; transcript_biotype "protein_coding";\n')
    bar = os.path.join(d, "bar.fa")
    with open(bar, "w") as f:
        for i in range(3):
            b = rnd(20)
            f.write(">%06d\n%s\n" % (100000 + i, b.lower() if i == 1 else b))
    with open(os.path.join(d, "genes.lst"), "w") as f:
        f.write("ENSG01\nENSG02 comment\nENSG03\nENSG04\n")
    custom = [("amp1", seqs[0][3000:3300].upper()), ("amp2", rnd(200))]
    with open(os.path.join(d, "custom.fa"), "w") as f:
        f.write(">amp1 x\n%s\n>amp2\n%s\n" % (custom[0][1], custom[1][1]))
    text = ("\n".join(s.upper() for s in seqs) + "\n").encode()
    return dict(dir=d, fa=fa, gtf=gtf, bar=bar, seqs=seqs, text=text, custom=custom, fm9=os.path.join(d, "GRCh38_toy.fa.fm9"),
                gtf_text=gzip.open(gtf, "rt").read(), bar_text=open(bar).read())


def oracle_run(orc, sc, case, out, js):
    """The oracle's (tsv, json, stderr, rc) for one case, echoing the same paths the binary was given."""
    label, args, inp, kw = case
    infile = os.path.join(sc["dir"], inp[1:]) if inp.startswith("@") else inp
    okw = dict(genome=sc["fa"], infile=infile, outfile=out, barcodes=sc["bar"], gtf=sc["gtf"], jsonfile=js, json=True, ucsc="hg38")
    okw.update(kw)
    if okw.get("input_fasta"):
        return orc.padlock([n for n, _ in sc["custom"]], [s for _, s in sc["custom"]], "", sc["bar_text"], **okw)
    return orc.padlock(NAMES, sc["seqs"], sc["gtf_text"], sc["bar_text"], **okw)
