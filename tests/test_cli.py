"""The `cammiq --query` shell: flag contract of the reference CLI (main.cpp:74-446), stderr
lines of query64_p (query.cpp:642-647) and the --read_cnts TSV (query.cpp:1786-1818)."""
import os
import subprocess

import pytest

from cammiq_amd import synth
from util import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "cammiq_amd", "cammiq")


def _run(args, cwd=None):
    return subprocess.run([CLI] + args, cwd=cwd, capture_output=True, text=True)


def test_cli_rejects_what_the_reference_rejects():
    assert os.path.exists(CLI), "cammiq shell not built"
    r = _run(["--query", "--bogus"])
    assert r.returncode != 0 and "Failed to recognize option: --bogus." in r.stderr
    r = _run(["--read_cnts"])
    assert r.returncode != 0 and "only valid in mode QUERY" in r.stderr
    r = _run(["--query", "-h", "40", "-i", "x.bin1"])
    assert r.returncode != 0 and "range [5, 31]" in r.stderr
    r = _run(["--query", "-f", "m.out", "-i", "x.bin1"])
    assert r.returncode != 0 and "Please specify at least one query file or directory." in r.stderr
    r = _run(["--build", "-k", "26"])
    assert r.returncode != 0 and "out of scope" in r.stderr


@pytest.mark.gpu
def test_cli_read_cnts_matches_reference_tsv(tmp_path):
    """survey_F1: the reference CLI's own TSV for this input is recorded in SURVEY.md 8(c)
    (q1.fastq 412 317 404 417 under taxids 1001..1004)."""
    g = golden("survey_F1")
    fq = tmp_path / "q1.fastq"
    synth.write_fastq(str(fq), g["reads"])
    out = tmp_path / "out_sc.txt"
    r = _run(["--query", "--read_cnts", "-f", os.path.join(g["dir"], "genome_map.out"), "-i", g["pu"], g["pd"],
              "-q", str(fq), "-o", str(out), "-t", "1"])
    assert r.returncode == 0, r.stderr
    assert out.read_text() == "QUERY/TAXID\t1001\t1002\t1003\t1004\nq1.fastq\t412\t317\t404\t417\n"
    assert "Number of unlabeled reads: 263." in r.stderr
    assert "Number of reads with conflict labels: 0." in r.stderr
    assert "Time for query:" in r.stderr and "Querying q1.fastq." in r.stderr


@pytest.mark.gpu
def test_cli_quantification_mode_and_dump(tmp_path):
    g = golden("f_deep")
    fq = tmp_path / "a.fq"
    synth.write_fastq(str(fq), g["reads"])
    dump = tmp_path / "counts.tsv"
    r = _run(["--query", "-f", os.path.join(g["dir"], "genome_map.out"), "-i", g["pu"], g["pd"], "-q", str(fq),
              "-t", "4", "--dump_counts", str(dump)])
    assert r.returncode == 0, r.stderr
    e = g["exp"]["p"]
    assert f"Number of unlabeled reads: {e['nundet']}." in r.stderr
    assert f"Number of reads with conflict labels: {e['nconf']}." in r.stderr
    assert "Loaded genome length file." in r.stderr
    rows = [l.split("\t") for l in dump.read_text().splitlines()]
    assert rows[0] == ["#cammiq_counts", "2"]
    assert rows[1][:2] == ["#query", "a.fq"] and int(rows[1][3]) == len(g["reads"])
    assert int(rows[1][5]) == sum(len(x) for x in g["reads"])
    gl = [x for x in rows if x[0] == "G"]
    assert [int(x[3]) for x in gl] == e["cnt_u"][1:] and [int(x[4]) for x in gl] == e["cnt_d"][1:]
    # a13: glength / nus / nds of every genome, as the three text files next to index_u give them
    meta = {k: dict(tuple(map(int, l.split())) for l in open(os.path.join(g["dir"], fn)))
            for k, fn in (("gl", "genome_lengths.out"), ("nu", "unique_lmer_count_u.out"), ("nd", "unique_lmer_count_d.out"))}
    for x in gl:
        i = int(x[1])
        assert (int(x[5]), int(x[6]), int(x[7])) == (meta["gl"][i], meta["nu"][i], meta["nd"].get(i, 0))
    ru = {int(x[2]): int(x[8]) for x in rows if x[0] == "L" and x[1] == "u"}
    assert ru == {i: v for i, v in enumerate(e["rcount_u"]) if v}
    rd = {int(x[2]): int(x[8]) for x in rows if x[0] == "L" and x[1] == "d"}
    assert rd == {i: v for i, v in enumerate(e["rcount_d"]) if v}


def _copy_fixture(g, dst, skip=()):
    import shutil
    for f in os.listdir(g["dir"]):
        if f not in skip and f != "reads.txt.gz" and f != "expected.json":
            shutil.copy(os.path.join(g["dir"], f), os.path.join(dst, f))
    return os.path.join(dst, "index_u.bin1"), os.path.join(dst, "index_d.bin2")


@pytest.mark.parametrize("missing,msg", [("genome_lengths.out", "Can not open genome length file."),
                                         ("unique_lmer_count_u.out", "Can not open unique count file."),
                                         ("unique_lmer_count_d.out", "Can not open doubly-unique count file.")])
def test_cli_fails_like_the_reference_when_a_meta_file_is_missing(tmp_path, missing, msg):
    """FqReader::loadGenomeLength (query.cpp:158-205) aborts when one of the three text files next to
    index_u cannot be opened; the shell exits non-zero with the same message.  Runs without a GPU:
    the index is loaded on the host first, the files are checked before any device work is needed."""
    g = golden("survey_F1")
    pu, pd = _copy_fixture(g, str(tmp_path), skip=(missing,))
    fq = tmp_path / "q.fastq"
    synth.write_fastq(str(fq), g["reads"][:10])
    r = _run(["--query", "-f", str(tmp_path / "genome_map.out"), "-i", pu, pd, "-q", str(fq), "--device", "-1"])
    assert r.returncode != 0 and msg in r.stderr, r.stderr


def test_cli_meta_rules_follow_the_reference(tmp_path):
    """--read_cnts with -Q does not read the meta files (query.cpp:303-331), with -q it does (:347);
    a --unique index (no .bin2) may lack unique_lmer_count_d.out (never written, build.cpp:671-698)."""
    g = golden("f_flat")    # unique-only fixture
    pu, _ = _copy_fixture(g, str(tmp_path), skip=("unique_lmer_count_d.out",))
    qd = tmp_path / "q"
    qd.mkdir()
    synth.write_fastq(str(qd / "reads.fastq"), g["reads"][:10])
    common = ["--query", "-f", str(tmp_path / "genome_map.out"), "-i", pu, "--device", "-1"]
    # host-only handle: everything up to the classify call runs, which then reports "no device"
    r = _run(common + ["-q", str(qd / "reads.fastq")])
    assert "Loaded genome length file." in r.stderr and "host-only" in r.stderr
    os.remove(tmp_path / "unique_lmer_count_u.out")
    r = _run(common + ["--read_cnts", "-Q", str(qd) + "/"])
    assert "Can not open" not in r.stderr and "host-only" in r.stderr
    r = _run(common + ["--read_cnts", "-q", str(qd / "reads.fastq")])
    assert r.returncode != 0 and "Can not open unique count file." in r.stderr


def _py_fastq_digest(path, min_l=0):
    """Independent parse: second line of every four, CR stripped, length filter, one
    deterministic substitute base per read for all its N's (cammiq_main.cpp read_fastq)."""
    M = (1 << 64) - 1
    bases = bytearray()
    offs = [0]
    for rec, chunk in enumerate(_records(path)):
        seq = chunk
        if len(seq) < min_l:
            continue
        if b"N" in seq:
            z = ((rec + 1) * 0x9E3779B97F4A7C15) & M
            z ^= z >> 29
            seq = seq.replace(b"N", b"ACGT"[(z >> 7) & 3:((z >> 7) & 3) + 1])
        bases += seq
        offs.append(len(bases))
    h = 1469598103934665603
    for c in bases:
        h = ((h ^ c) * 1099511628211) & M
    for x in offs:
        h = ((h ^ x) * 1099511628211) & M
    return len(offs) - 1, len(bases), h


def _records(path):
    lines = open(path, "rb").read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    for i in range(1, len(lines), 4):
        yield lines[i].rstrip(b"\r")


@pytest.mark.parametrize("n_reads,crlf,final_newline", [(50, False, True), (50, True, False), (40000, False, True),
                                                        (40000, True, False)])
def test_fastq_loader_matches_independent_parse(tmp_path, n_reads, crlf, final_newline):
    """Small files take the single-thread path, > 4 MiB files the multi-threaded mmap path."""
    import numpy as np
    rng = np.random.default_rng(n_reads + crlf)
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n_reads):
        L = int(rng.integers(20, 250))
        s = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, size=L, p=[.245, .245, .245, .245, .02])].tobytes()
        out.append(b"@r%d" % i + nl + s + nl + b"+" + nl + b"I" * L)
    data = nl.join(out) + (nl if final_newline else b"")
    fq = tmp_path / "x.fastq"
    fq.write_bytes(data)
    for min_l in (0, 100):
        r = _run(["--query", "--read_length_filter", str(min_l), "--fastq_stats", str(fq)])
        assert r.returncode == 0, r.stderr
        n, b, h = _py_fastq_digest(str(fq), min_l)
        assert r.stdout.split() == ["reads", str(n), "bases", str(b), "fnv", "%016x" % h]


@pytest.mark.gpu
def test_cli_several_fastq_files(tmp_path):
    """Three query files in one run: the next file is parsed while the previous one is classified;
    every file gets its own TSV row and its own counters (the reference resets them per file,
    query.cpp:1820-1840), equal to what each file gives alone."""
    import cammiq_amd as cq
    g = golden("survey_F1")
    reads = g["reads"]
    parts = [reads[:700], reads[700:701], reads[701:]]
    names = []
    for i, part in enumerate(parts):
        fq = tmp_path / f"part{i}.fastq"
        synth.write_fastq(str(fq), part)
        names.append(str(fq))
    out = tmp_path / "out.txt"
    r = _run(["--query", "--read_cnts", "-f", os.path.join(g["dir"], "genome_map.out"), "-i", g["pu"], g["pd"],
              "-q"] + names + ["-o", str(out)])
    assert r.returncode == 0, r.stderr
    lines = out.read_text().splitlines()
    assert lines[0] == "QUERY/TAXID\t1001\t1002\t1003\t1004" and len(lines) == 4
    ix = cq.Index(g["pu"], g["pd"], device=0)
    total = [0, 0, 0, 0]
    for i, part in enumerate(parts):
        b, o = synth.concat_reads(part)
        want = [int(x) for x in ix.query(b, o, g["G"], mode=cq.MODE_SC)["cnt_u"][1:]]
        got = lines[1 + i].split("\t")
        assert got[0] == f"part{i}.fastq" and [int(x) for x in got[1:]] == want
        total = [a + c for a, c in zip(total, want)]
    assert total == [412, 317, 404, 417]          # the whole sample
    assert r.stderr.count("Time for query:") == 3 and r.stderr.count("Loaded query file") == 3


@pytest.mark.gpu
def test_cli_packs_fastq_straight_to_two_bit_rows(tmp_path):
    """Queries never materialise the ASCII reads: pass 2 of the FASTQ loader packs every sequence line with
    cq_pack_read.  Reads with N (one substitute base per read, derived from the record number), lower case, CR LF,
    reads shorter than h and longer than 255 bases: the TSV must equal cq_query on the substituted reads."""
    import numpy as np
    import cammiq_amd as cq
    g = golden("survey_F1")
    M = (1 << 64) - 1
    rng = np.random.default_rng(5)
    reads = []
    for i, r in enumerate(g["reads"][:1500]):
        r = bytearray(r)
        if i % 7 == 0:
            for j in rng.choice(len(r), 2, replace=False):
                r[j] = ord("N")
        if i % 11 == 0:
            r = bytearray(bytes(r).lower().replace(b"n", b"N"))
        reads.append(bytes(r))
    reads += [b"ACGT", g["reads"][3] * 4, b"ACGTNACGT" * 3]                      # too short, too long (> 255), short with N
    fq = tmp_path / "n.fastq"
    fq.write_bytes(b"".join(b"@r%d\r\n%s\r\n+\r\n%s\r\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads)))
    out = tmp_path / "out.txt"
    r = _run(["--query", "--read_cnts", "-f", os.path.join(g["dir"], "genome_map.out"), "-i", g["pu"], g["pd"],
              "-q", str(fq), "-o", str(out)])
    assert r.returncode == 0, r.stderr
    sub = []
    for rec, s in enumerate(reads):
        if b"N" in s:
            z = ((rec + 1) * 0x9E3779B97F4A7C15) & M
            z ^= z >> 29
            s = s.replace(b"N", b"ACGT"[(z >> 7) & 3:((z >> 7) & 3) + 1])
        sub.append(s)
    b, o = synth.concat_reads(sub)
    want = cq.Index(g["pu"], g["pd"], device=0).query(b, o, g["G"], mode=cq.MODE_SC)
    got = [int(x) for x in out.read_text().splitlines()[1].split("\t")[1:]]
    assert got == [int(x) for x in want["cnt_u"][1:]]
    assert f"Number of unlabeled reads: {want['nundet']}." in r.stderr
    assert want["nskipped"] == 2 and "(skipped): 2." in r.stderr
