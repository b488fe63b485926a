"""Where the first sight of a dialogue geometry spends its time in beam_search_decode (cProfile of one turn on an unseen geometry;
development aid)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
import bist_amd.model.decode as D
from bist_amd.data.synthetic import synthetic_batch

c = bench.CFG
args = bench.model_args(6, 512, 8, 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda().to(torch.bfloat16).eval()


def turn(Lq, Lh, Lc):
    b1 = synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=Lq, Lh=Lh, Lc=Lc, Lt=c["Lt"], vocab=c["V"], seed=Lq, dtype=torch.bfloat16)
    with torch.no_grad():
        D.beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
    torch.cuda.synchronize()


turn(24, 64, 32); turn(16, 48, 24)
pr = cProfile.Profile()
pr.enable(); turn(8, 40, 16); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
