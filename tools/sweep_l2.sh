for cfg in "16 1000 1024" "32 1000 1024" "16 8000 4096" "20 8000 4096" "32 8000 4096" "16 4000 1024" "12 2000 1024" "12 8000 1024" "14 8000 1024" "24 20000 1024" "32 40000 1024" "16 20000 1024"; do
  for t in none l2=1; do SMH_WM_TUNE=$t timeout -k 10 60 python tools/wmbench.py $cfg 4 2>&1 | grep "^WM" | sed "s/^/[$t] /"; done
done
