"""Where does the product leave the reference's trajectory?  The first epoch of synth.SOLVER_RUN step by step: the product's
Solver.sup_train_one_iteration on the GPU next to the oracle's sup_train_step on the host (the oracle reproduces the
reference to 1e-6), then the dev loss of (a) the product, (b) the oracle on ITS weights, (c) the oracle on the PRODUCT's weights -
which separates "the weights have drifted" from "the evaluation path differs"."""
import contextlib, io, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + "/semi-supervised-asr_amd", ROOT + "/tests/golden", ROOT + "/tests"]
import numpy as np, torch, yaml
import __graft_entry__ as entry
entry.build()
import hip_backend as hb
import synth
from oracle import asr_oracle as O
from solver import Solver
hb.ARITH[0] = hb.ARITH_NAMES[os.environ.get("ARITH", "bf16x6")]
torch.set_num_threads(16)
root = tempfile.mkdtemp(); os.chdir(root)
run = synth.SOLVER_RUN
base = yaml.safe_load(open(ROOT + "/semi-supervised-asr_amd/config.yaml"))
synth.write_solver_run_corpus(root)
cfg = synth.solver_run_config(base, root)
with contextlib.redirect_stdout(io.StringIO()):
    s = Solver(cfg)
mcfg, _ = synth.solver_run_model_cfg(cfg)
w = synth.e2e_weights(mcfg, run["model_wseed"])
s.model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
sd = O.make_leaf_state(w)
ocfg = dict(mcfg, dropout_rate=0.0, ls_weight=cfg["ls_weight"], labeldist=s.labeldist)
names = O.unique_param_names(sd)
opt = O.AdamAmsgrad(names, lr=cfg["learning_rate"], weight_decay=cfg["weight_decay"])
dev = torch.device("cuda", 0)

def wdist():
    psd = s.model.state_dict()
    num = sum(float(((psd[n].cpu().double() - sd[n].detach().double()) ** 2).sum()) for n in names)
    den = sum(float((sd[n].detach().double() ** 2).sum()) for n in names)
    return (num / den) ** 0.5

np.random.seed(run["numpy_seed"])
for it, (xs, ilens, ys) in enumerate(s.train_lab_loader):
    st = np.random.get_state()
    lp = float(s.sup_train_one_iteration(xs.to(dev), ilens, [y.to(dev) for y in ys], 1.0))
    np.random.set_state(st)
    lo, gn, _ = O.sup_train_step(sd, ocfg, opt, xs, ilens, ys, tf_rate=1.0, max_grad_norm=cfg["max_grad_norm"])
    if it % 5 == 0 or it == len(s.train_lab_loader) - 1:
        print("step %2d loss product %.6f oracle %.6f rel %.1e | gnorm %.3f | weights rel dist %.2e" % (it, lp, lo, abs(lp - lo) / lo, gn, wdist()))
s.flush()
s.model.eval()
tot = [0.0, 0.0, 0.0]
psd = O.make_leaf_state({k: v.detach().cpu().numpy() for k, v in s.model.state_dict().items()})
n = 0
for xs, ilens, ys in s.dev_loader:
    with torch.no_grad():
        st = np.random.get_state()
        _, lp, _, _ = s.model(xs.to(dev), ilens, ys=[y.to(dev) for y in ys])
        a = float(s.model.mask_and_cal_loss(lp, [y.to(dev) for y in ys]))
        np.random.set_state(st)
        _, lo, _, _ = O.e2e_forward(sd, ocfg, xs, ilens, ys, training=False)
        b = float(O.masked_loss(lo, ys))
        np.random.set_state(st)
        _, lq, _, _ = O.e2e_forward(psd, ocfg, xs, ilens, ys, training=False)
        c = float(O.masked_loss(lq, ys))
    tot[0] += a; tot[1] += b; tot[2] += c; n += 1
print("dev loss over %d batches: product %.6f | oracle on its own weights %.6f | oracle on the product's weights %.6f" % (n, tot[0] / n, tot[1] / n, tot[2] / n))
