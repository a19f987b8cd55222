import sys, time, os
sys.path.insert(0,'cuda-aho-corasick-wu-manber_amd'); sys.path.insert(0,'tests')
import numpy as np, smatcher_hip as S
n = 1<<30
text = S.corpus_text(n, 42, 4)
pat = S.corpus_patterns(8, 1000, 7, 4, 42, n, 2)
ac = S.AcAutomaton.from_patterns(pat, 8, 1000, 4)
for rep in range(4):
    t0=time.perf_counter(); c,ks = ac.count_host(text); dt=time.perf_counter()-t0
    print("count_host 1 GiB: %.1f GB/s (%.4f s), kernel %.4f s, count %d" % (n/dt/1e9, dt, ks, c), flush=True)
# fresh buffers: what a caller that hands over a new array every time sees
for rep in range(3):
    t2 = text.copy()
    t0=time.perf_counter(); c2,ks = ac.count_host(t2); dt=time.perf_counter()-t0
    print("fresh buffer: %.1f GB/s count ok %s" % (n/dt/1e9, c2==c), flush=True)
os.environ["SMH_HOST_PIECE_KIB"]="16384"
t0=time.perf_counter(); c3,ks = ac.count_host(text); dt=time.perf_counter()-t0
print("16 MiB pieces: %.1f GB/s ok %s" % (n/dt/1e9, c3==c))
os.environ["SMH_HOST_PIECE_KIB"]="4"
c4,ks = ac.count_host(text[:1<<22]); 
del os.environ["SMH_HOST_PIECE_KIB"]
c5,ks = ac.count_host(text[:1<<22])
print("4 KiB pieces on 4 MiB:", c4, c5, c4==c5)
S.lib.smh_host_path_release()
