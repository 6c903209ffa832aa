import sys, time, numpy as np, torch
sys.path.insert(0,'rust-triplets_amd'); sys.path.insert(0,'tests')
import triplets_amd as T, oracle_lib as O, textgen
ctx=T.Context(0)
den=T.DenoiserConfig(True,0.35,True); st=T.ChunkingStrategy()
for nrec in (1, 16, 256, 4096):
    recs=textgen.corpus(5, nrec)
    data,offs=O.pack(recs)
    db=torch.from_numpy(np.concatenate([data,np.zeros(64,np.uint8)])).cuda(); do=torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    for _ in range(20): ctx.preprocess_batch_device(db.data_ptr(),do.data_ptr(),len(recs),len(data),den,st)
    t0=time.perf_counter()
    for _ in range(200): ctx.preprocess_batch_device(db.data_ptr(),do.data_ptr(),len(recs),len(data),den,st)
    dt=(time.perf_counter()-t0)/200
    print("records %5d bytes %8d: %.1f us per call"%(nrec,len(data),dt*1e6), flush=True)
ctx.set_profiling(True)
ctx.preprocess_batch_device(db.data_ptr(),do.data_ptr(),len(recs),len(data),den,st)
print(ctx.profile())
