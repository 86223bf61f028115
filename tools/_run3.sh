set -x
mkdir -p gpurun_out/r3c
nproc; free -g | head -2; df -h /dev/shm | tail -1
python tools/configs4.py --genomes 15000 --genome-len 100000 --reads 2000000 --out gpurun_out/r3c/cfg4_debug.json > gpurun_out/r3c/cfg4_debug.log 2>&1 || { tail -30 gpurun_out/r3c/cfg4_debug.log; exit 1; }
grep "configs4\]" gpurun_out/r3c/cfg4_debug.log | tail -12
python tools/configs4.py --genomes 5000 --genome-len 3450000 --reads 20000000 --out gpurun_out/r3c/cfg4_mid.json > gpurun_out/r3c/cfg4_mid.log 2>&1 || { tail -30 gpurun_out/r3c/cfg4_mid.log; exit 1; }
grep -v "^{" gpurun_out/r3c/cfg4_mid.log | tail -60
