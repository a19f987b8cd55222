"""Development aid: compile / scan / free cycles of every handle type; device and host memory must stay flat."""
import sys, os
sys.path.insert(0,'/root/repo/cuda-aho-corasick-wu-manber_amd')
import torch, numpy as np
import smatcher_hip as S
text=S.corpus_text(1<<22,42,4)
free0,_=torch.cuda.mem_get_info()
import resource
r0=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for it in range(600):
    if it % 100 == 0:
        torch.cuda.synchronize(); print(it, 'device free %.1f MiB, host rss %d MiB' % (torch.cuda.mem_get_info()[0]/2**20, int(open('/proc/self/statm').read().split()[1])*4096>>20), flush=True)
    m=int(np.random.randint(3,33)); p=int(np.random.randint(1,2000))
    pat=S.corpus_patterns(m,p,7+it,4,42,1<<22,2)
    for cls in (S.AcAutomaton, S.WmTables, S.ShTrie, S.SbomOracle):
        try:
            h=cls.from_patterns(pat,m,p,4)
        except S.SmhError:
            continue
        h.count_host(text)
        h.close()
    ln=np.random.randint(3,12,size=20).astype(np.uint32)
    ps=S.PatternSet(np.random.randint(0,4,size=int(ln.sum())).astype(np.uint8), ln, 4, it%2)
    ps.count_host(text); ps.close()
    # wide DNA sets: grouped pair-gram filter / automaton / (every 50th: thousands of patterns) the split form
    npat = 2400 if it % 50 == 49 else 120
    ln=np.random.randint(8 if it % 3 else 14,41,size=npat).astype(np.uint32)
    ps=S.PatternSet(np.random.randint(0,4,size=int(ln.sum())).astype(np.uint8), ln, 4, S.ALGO_WM)
    ps.count_host(text); ps.close()
torch.cuda.synchronize()
free1,_=torch.cuda.mem_get_info()
r1=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("device free before %.1f MiB after %.1f MiB; host maxrss %d -> %d MiB" % (free0/2**20, free1/2**20, r0>>10, r1>>10))
