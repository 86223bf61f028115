"""Index load time on the configs[1] index (500 genomes, --unique): decode + layout vs the
CAMMIQ_IMAGE_CACHE file.  Run on the GPU box:  python tools/load_time.py [--both]"""
import argparse
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cammiq_amd as cq
from cammiq_amd import bigsynth

ap = argparse.ArgumentParser()
ap.add_argument("--genomes", type=int, default=500)
ap.add_argument("--genome-len", type=int, default=3_450_000)
ap.add_argument("--both", action="store_true")
ap.add_argument("--device", type=int, default=0)
a = ap.parse_args()

wdir = f"/dev/shm/cammiq_loadtime_{os.getpid()}"
os.makedirs(wdir)
try:
    w = bigsynth.World(seed=2, n_genomes=a.genomes, genome_len=a.genome_len, k=26, h=26, lmax=50,
                       frac_deep=0.07, pair_share=0.3 if a.both else 0.0)
    pu = os.path.join(wdir, "index_u.bin1")
    pd = os.path.join(wdir, "index_d.bin2") if a.both else None
    w.write_index(pu, pd)
    out = {"genomes": a.genomes, "both": a.both, "index_bytes": sum(
        os.path.getsize(os.path.join(wdir, f)) for f in os.listdir(wdir))}

    def load():
        t = time.time()
        ix = cq.Index(pu, pd, device=a.device)
        dt = time.time() - t
        info = ix.info_dict()
        cached = ix.info.reserved_
        del ix
        return dt, cached, info
    out["decode_s"] = [round(load()[0], 3) for _ in range(2)]
    os.environ["CAMMIQ_IMAGE_CACHE"] = "force"   # "1" no longer caches for a GPU handle whose table the device lays out: that load is the faster one
    dt, c, info = load()
    assert c == 0
    out["decode_and_write_cache_s"] = round(dt, 3)
    out["cache_bytes"] = os.path.getsize(pu + ".cqimg")
    res = [load() for _ in range(2)]
    assert all(r[1] == 1 for r in res) and all(r[2] == info for r in res)
    out["cached_s"] = [round(r[0], 3) for r in res]
    print(json.dumps(out))
finally:
    shutil.rmtree(wdir, ignore_errors=True)
