# tools/sweep_forms.sh -- DNA sets x the three pair-like gram forms (compiled under "gram=1|3|5"), full scan and filter alone,
# interleaved per set (tools/gram_ab.py): what the cost model of wm_host.c (gram_verify_ms_pairlike) is fitted to
for cfg in "10 2000" "12 2000" "12 8000" "12 20000" "14 8000" "16 8000" "16 20000" "16 40000" "20 20000" "24 20000" "24 40000" "32 40000"; do
  timeout -k 10 120 python tools/gram_ab.py $cfg 1024 4 "gram=1 debug" "gram=3 debug" "gram=5 debug" 2>&1 | grep -v "amdgpu\|block filter est" 
done
