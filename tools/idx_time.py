import time, os, sys, tempfile
sys.path.insert(0, '/root/repo')
from monica_amd import _capi, synth
names, seqs = synth.genome_set(20, min_len=2_000_000, max_len=7_000_000)
t=time.time(); idx=_capi.Index.from_seqs(names, seqs); t1=time.time()-t
d=tempfile.mkdtemp(); p=os.path.join(d,'i.mmi')
t=time.time(); idx.save(p); t2=time.time()-t
t=time.time(); idx2=_capi.Index.load(p); t3=time.time()-t
t=time.time(); eng=_capi.Engine(idx2, 0); t4=time.time()-t
t=time.time(); eng2=_capi.Engine(idx2, 0); t5=time.time()-t
print("build %.2f s, save %.2f, load %.2f, first engine (upload + tables) %.2f, second engine %.3f; file %.0f MB" % (t1,t2,t3,t4,t5, os.path.getsize(p)/1e6))
names2, seqs2 = synth.genome_set(2, min_len=50_000, max_len=60_000)
small = _capi.Index.from_seqs(names2, seqs2)
t=time.time(); e3=_capi.Engine(small, 0); print("engine on a tiny index (constant part) %.3f s" % (time.time()-t))
