O=gpurun_out/r02_ax; mkdir -p $O
( timeout 120 python tools/wm_ab.py 32 8000 1024 4 x hd=1 2>&1 | grep -v "amdgpu\|in order"
  timeout 120 python tools/wm_ab.py 32 8000 1024 4 x hd=1,stmin=2 2>&1 | grep -v "amdgpu\|in order"
  timeout 120 python tools/wm_ab.py 32 8000 1024 4 x hd=1,stmin=4 2>&1 | grep -v "amdgpu\|in order"
  timeout 120 python tools/wm_ab.py 16 8000 1024 4 x hd=1 2>&1 | grep -v "amdgpu\|in order"
  timeout 120 python tools/wm_ab.py 16 8000 1024 4 x hd=1,stmin=4 2>&1 | grep -v "amdgpu\|in order"
  timeout 120 python tools/wm_ab.py 24 8000 1024 4 x hd=1 2>&1 | grep -v "amdgpu\|in order" ) > $O/bench.log 2>&1
cat $O/bench.log
