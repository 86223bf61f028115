set -x
mkdir -p gpurun_out/r3d
nproc; free -g | head -2
timeout -k 10 900 python tools/configs4.py --fit --genomes 15000 --genome-len 3450000 --reads 20000000 --out gpurun_out/r3d/cfg4_full.json > gpurun_out/r3d/cfg4_full.log 2>&1 || { tail -30 gpurun_out/r3d/cfg4_full.log; exit 1; }
grep -v "^{" gpurun_out/r3d/cfg4_full.log | tail -40
L=$(python3 -c "import json;print(json.load(open('gpurun_out/r3d/cfg4_full.json'))['genome_len'])")
MIB=$(python3 -c "import json;print(int(json.load(open('gpurun_out/r3d/cfg4_full.json'))['table']['table_GB']*1e9/1048576))")
timeout -k 10 200 tools/gather_bench $MIB > gpurun_out/r3d/gather_cfg4.txt 2>&1; cat gpurun_out/r3d/gather_cfg4.txt
for m in 18 17; do
  CAMMIQ_LIB=$PWD/variants/libcammiq_m$m.so timeout -k 10 600 python tools/configs4.py --genomes 15000 --genome-len $L --reads 20000000 --out gpurun_out/r3d/cfg4_full_m$m.json > gpurun_out/r3d/cfg4_full_m$m.log 2>&1 || { tail -30 gpurun_out/r3d/cfg4_full_m$m.log; exit 1; }
  grep -E "table \{|index_load|device queries" gpurun_out/r3d/cfg4_full_m$m.log
  python3 -c "import json;d=json.load(open('gpurun_out/r3d/cfg4_full_m$m.json'));print('m$m',d['kernel_ms_runs'],d['kernel_Gwindows_s'],d['checks'])"
done
