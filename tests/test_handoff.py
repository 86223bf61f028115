"""SURVEY.md 8(f) rank 3 -- the ILP hand-off.  include/cammiq_glue.hpp is the reference-side binding
(INTEGRATION.md); tests/cpp/ilp_handoff.cpp compiles it as C++11 against an own mock of the
reference's Genome / pleafNode / Hash and prints the state runILP_* would read
(/root/reference/src/query.cpp:1100-1226).  Here that state is checked against the oracle:
map_sp order = decode order, every doubly-unique leaf in both lists (hashtrie.cpp:452-453,476),
glength / nus / nds from the three text files (query.cpp:158-205), counters and rcount after a query."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from cammiq_amd import synth
from util import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def handoff_bin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("handoff") / "ilp_handoff")
    lib = os.path.join(ROOT, "cammiq_amd")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-o", out,
                           os.path.join(ROOT, "tests", "cpp", "ilp_handoff.cpp"), "-L" + lib, "-lcammiq_hip",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def _meta(d):
    rd = lambda fn: dict(tuple(map(int, l.split())) for l in open(os.path.join(d, fn))) if os.path.exists(os.path.join(d, fn)) else {}
    return rd("genome_lengths.out"), rd("unique_lmer_count_u.out"), rd("unique_lmer_count_d.out")


def _expected(g, counts=None):
    """The same text, from the oracle's decode-order leaves (+ counts of a query)."""
    oi = oracle_lib.OracleIndex(g["pu"], g["pd"])
    G = g["G"]
    gl, nu, nd = _meta(g["dir"])
    lv = [oi.leaves(0), oi.leaves(1)]
    rc = [counts["rcount_u"], counts["rcount_d"]] if counts else [np.zeros(oi.n_leaves[0], int), np.zeros(oi.n_leaves[1], int)]
    sp = [[[] for _ in range(G + 1)] for _ in (0, 1)]
    for t in (0, 1):
        for i in range(oi.n_leaves[t]):
            sp[t][int(lv[t]["refID1"][i])].append(i)
            if lv[t]["refID2"][i]:
                sp[t][int(lv[t]["refID2"][i])].append(i)
    out = [f"nundet {counts['nundet'] if counts else 0} nconf {counts['nconf'] if counts else 0}"]
    for i in range(1, G + 1):
        cu = int(counts["cnt_u"][i]) if counts else 0
        cd = int(counts["cnt_d"][i]) if counts else 0
        out.append(f"G {i} {cu} {cd} {gl.get(i, 0)} {nu.get(i, 0)} {nd.get(i, 0)} {len(sp[0][i])} {len(sp[1][i])}")
        for t in (0, 1):
            for k in sp[t][i]:
                L = lv[t]
                out.append(f"{'ud'[t]} {k} {L['refID1'][k]} {L['refID2'][k]} {L['depth'][k]} {L['ucount1'][k]} "
                           f"{L['ucount2'][k]} {int(rc[t][k])}")
    for (a, b), c in sorted((counts or {}).get("pairs", {}).items()):
        out.append(f"P {a} {b} {c}")
    out.append(f"leaf_cnt {oi.n_leaves[0]} {oi.n_leaves[1]}")
    return out


@pytest.mark.parametrize("name", ["survey_F1", "survey_F2", "f_deep", "f_flat"])
def test_map_sp_and_meta_without_a_gpu(handoff_bin, name):
    g = golden(name)
    r = subprocess.run([handoff_bin, g["pu"], g["pd"] or "-", str(g["G"]), "-1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == _expected(g)


def test_missing_meta_file_is_reported_with_the_reference_message(handoff_bin, tmp_path):
    import shutil
    g = golden("survey_F1")
    for f in os.listdir(g["dir"]):
        if f.startswith("index_") or f == "genome_lengths.out":
            shutil.copy(os.path.join(g["dir"], f), tmp_path / f)
    r = subprocess.run([handoff_bin, str(tmp_path / "index_u.bin1"), str(tmp_path / "index_d.bin2"), "4", "-1"],
                       capture_output=True, text=True)
    assert r.returncode == 3 and r.stderr == "Can not open unique count file.\n"


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", [("f_deep", "P"), ("survey_F2", "P"), ("f_deep", "SC")])
def test_state_after_a_gpu_query_equals_the_oracle(handoff_bin, tmp_path, name, mode):
    g = golden(name)
    rf = tmp_path / "reads.txt"
    rf.write_bytes(b"\n".join(g["reads"]) + b"\n")
    r = subprocess.run([handoff_bin, g["pu"], g["pd"] or "-", str(g["G"]), "0", str(rf), mode], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    b, o = synth.concat_reads(g["reads"])
    ref = oracle_lib.OracleIndex(g["pu"], g["pd"]).query(b, o, g["G"], mode=1 if mode == "SC" else 0)
    if mode == "SC":     # query64_sc leaves rcount alone
        ref["rcount_u"][:] = 0
        ref["rcount_d"][:] = 0
    else:
        ref["pairs"] = {}
    assert r.stdout.splitlines() == _expected(g, ref)
