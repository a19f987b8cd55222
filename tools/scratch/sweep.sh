for m in 12 8 20 5; do for p in 100000 85000 75000 60000 40000; do timeout -k 10 120 python tools/wmbench.py $m $p 4096 256 || exit 1; done; done
