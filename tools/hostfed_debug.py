#!/usr/bin/env python3
"""tools/hostfed_debug.py -- which knob of the host-fed pipeline changes the counts?  One index, one batch; the truth is the
device door (cq_query_device on HBM-resident rows); every variant of the host-fed door is compared with it field by field."""
import os, sys, json, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cammiq_amd as cq
from cammiq_amd import bigsynth

n, rl = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000, 100
G = int(sys.argv[2]) if len(sys.argv) > 2 else 300
glen = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
w = bigsynth.World(seed=2, n_genomes=G, genome_len=glen, pair_share=0.3)
wdir = tempfile.mkdtemp(dir="/dev/shm")
pu, pd = os.path.join(wdir, "u.bin1"), os.path.join(wdir, "d.bin2")
nu, nd = w.write_index(pu, pd)
ix = cq.Index(pu, pd, device=0)
shutil.rmtree(wdir)
sb, sw = cq.stride_bytes(rl), cq.stride_words(rl)
hp = cq.host_array(n * sb, np.uint8).reshape(n, sb); hl = cq.host_array(n, np.uint8)
dp = torch.empty((n, sw), dtype=torch.int32, device="cuda"); dl = torch.empty(n, dtype=torch.uint8, device="cuda")
buf = np.empty(4_000_000 * rl, np.uint8)
for c0 in range(0, n, 4_000_000):
    m = min(4_000_000, n - c0)
    w.reads_into(buf, 1000, c0, m, rl)
    offs = np.arange(m + 1, dtype=np.uint64) * np.uint64(rl)
    pk, ln, _ = cq.pack_reads(buf[:m * rl], offs, 26, sw)
    cq.pack_reads_tight(buf[:m * rl], offs, 26, sb, out=(hp[c0:c0 + m], hl[c0:c0 + m]))
    dp[c0:c0 + m].copy_(torch.from_numpy(pk.view(np.int32))); dl[c0:c0 + m].copy_(torch.from_numpy(ln))
cw = ix.counter_words(G)
ctr = torch.zeros(cw, dtype=torch.int64, device="cuda"); rc = torch.zeros(nu + nd, dtype=torch.int32, device="cuda")
ix.query_device(cq.MODE_P, dp.data_ptr(), dl.data_ptr(), n, sw, rl, G, ctr.data_ptr(), rc.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
c = ctr.cpu().numpy().astype(np.uint64); r = rc.cpu().numpy().view(np.uint32)
truth = dict(cnt_u=c[:G + 1], cnt_d=c[G + 1:2 * G + 2], nundet=int(c[2 * G + 2]), nconf=int(c[2 * G + 3]), rcount_u=r[:nu], rcount_d=r[nu:])
print("truth", int(truth["cnt_u"].sum()), int(truth["cnt_d"].sum()), truth["nundet"], truth["nconf"], int(truth["rcount_u"].astype(np.uint64).sum()), flush=True)
out = ix.counts_out(G, pinned=True)
knobs = ["CAMMIQ_TWO_STREAMS", "CAMMIQ_EARLY_NARROW", "CAMMIQ_RCOUNT_NARROW", "CAMMIQ_LENS_FILL", "CAMMIQ_CHUNK_TAIL", "CAMMIQ_NARROW_WGS"]
variants = [("default", {}), ("one_stream", {"CAMMIQ_TWO_STREAMS": "0"}), ("late_narrow", {"CAMMIQ_EARLY_NARROW": "0"}),
            ("one_stream+late_narrow", {"CAMMIQ_TWO_STREAMS": "0", "CAMMIQ_EARLY_NARROW": "0"}), ("narrow_off", {"CAMMIQ_RCOUNT_NARROW": "0"}),
            ("narrow_off+one_stream", {"CAMMIQ_RCOUNT_NARROW": "0", "CAMMIQ_TWO_STREAMS": "0"}), ("lens_copy", {"CAMMIQ_LENS_FILL": "0"}),
            ("default again", {})]
for name, env in variants:
    for k in knobs:
        os.environ.pop(k, None)
    os.environ.update(env)
    for rep in range(2):
        out.cu[:] = 7; out.cd[:] = 7
        q = ix.query_packed_tight(hp, hl, rl, G, out=out)
        bad = [k for k in ("cnt_u", "cnt_d", "rcount_u", "rcount_d") if not np.array_equal(np.asarray(q[k]).astype(np.uint64), np.asarray(truth[k]).astype(np.uint64))]
        bad += [k for k in ("nundet", "nconf") if q[k] != truth[k]]
        print(f"{name:26s} rep {rep}: {'EQUAL' if not bad else 'DIFFERS in ' + ','.join(bad)}  sums {int(q['cnt_u'].sum())} {int(q['cnt_d'].sum())} {q['nundet']} {q['nconf']} nskipped {q['nskipped']}", flush=True)
