import sys
sys.path.insert(0,'cuda-aho-corasick-wu-manber_amd')
import torch, numpy as np, smatcher_hip as S, time
dev=torch.device('cuda',0)
for kind, sig in ((1,4),(2,20),(2,256),(3,4),(3,256)):
    for off, n in ((0, 1<<20), (1024*5+16, 300000), (777, 70001), (1<<32, 1<<18)):
        t = torch.zeros(n+64, dtype=torch.uint8, device=dev)
        S.corpus_text_device(t.data_ptr(), n, 42, sig, off, kind)
        torch.cuda.synchronize()
        h = S.corpus_text(n, 42, sig, off, kind)
        assert np.array_equal(t[:n].cpu().numpy(), h), (kind, sig, off, n)
        assert int(t[n:].sum()) == 0
    t = torch.zeros((1<<30)+64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(); t0=time.time()
    S.corpus_text_device(t.data_ptr(), 1<<30, 42, sig, 0, kind); torch.cuda.synchronize()
    print("kind", kind, "alphabet", sig, "device == host; 1 GiB generated in %.1f ms" % ((time.time()-t0)*1e3))
