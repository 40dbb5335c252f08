import os, sys, socket, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.multiprocessing as mp
from test_gpu_parity import _two_rank_worker
if __name__ == "__main__":
    d = tempfile.mkdtemp()
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(_two_rank_worker, args=(2, port, d), nprocs=2, join=True, start_method="spawn")
    r0, r1 = (np.load(os.path.join(d, f"r{r}.npz")) for r in range(2))
    print("cand equal", np.array_equal(r0["cand"], r1["cand"]))
    df = np.abs(r0["particles"] - r1["particles"])
    print("max particle diff between ranks", df.max(), "at", np.argmax(df), "n differing", (df > 0).sum(), "of", df.size)
    P = 37
    print("differing particle ids", sorted(set((np.where(df > 0)[0] % P).tolist())))
