# same-box A/B of the register prefetch: the committed library (SMH_PREFETCH=0), then the kernels rebuilt with =1 on the box
O=gpurun_out/r02_ba; mkdir -p $O
run() {
  for cfg in "8 10000 1024 4" "16 1000 1024 4" "32 1000 1024 4" "32 8000 1024 4" "12 100000 1024 256" "20 100000 1024 256" "5 100000 1024 256"; do timeout 120 python tools/wmbench.py $cfg 2>&1 | grep -v amdgpu; done
  for cfg in "8 1000 1024" "16 1000 1024" "32 1000 1024" "8 8000 1024"; do timeout 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu; done
  timeout 300 python bench.py --no-cpu --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], 'm8/16/32', d['ac']['m8']['kernel_ms'], d['ac']['m16']['kernel_ms'], d['ac']['m32']['kernel_ms'], 'wm', d['wm']['kernel_ms'], 'wm_long', d['wm_long']['m16']['kernel_ms'], d['wm_long']['m32']['kernel_ms'], 'mixed', d['mixed_8_32']['ac']['kernel_ms'], 'ac8000', [d['ac_8000_patterns'][k]['kernel_ms'] for k in ('m8','m16','m32')], 'ascii', [d['wm_ascii'][k]['kernel_ms'] for k in ('m5','m12','m20')], 'stream', d['stream_read']['ms'], 'pos', d['positions']['kernel_ms'])"
}
( echo "=== SMH_PREFETCH=0 (committed build)"; run
  touch cuda-aho-corasick-wu-manber_amd/csrc/lane_common.h
  make -s -j16 -C cuda-aho-corasick-wu-manber_amd all HIPFLAGS='-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -DSMH_PREFETCH=1' > $O/make.log 2>&1; echo "make rc $?"
  echo "=== SMH_PREFETCH=1 (rebuilt here)"; run ) > $O/ab.log 2>&1
cat $O/ab.log
