"""Which bf16 rounding point costs what against the fp32 reference: the oracle's forward / backward with each of the
bf16 path's rounding points (weights, stored pre-BatchNorm tensors, stored activations, stored gradients) switched on
alone and together, and with the pre-BatchNorm tensors stored as deviations from a per-channel constant
("shift").  CPU only (test infrastructure: imports the oracle).  python tools_rounding_ablation.py [batch]
-> the JSON file named on the command line (committed under profiles/)"""
import sys, math, json
import os
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from oracle import vae_oracle as O

def fwd(x, sd, eps, rw, ry, ra, rg, shift=False, train=True):
    rf = (lambda t: O._rf(t))
    def R(t, on):
        if not on: return t
        t = O._rf(t)
        return O._rb(t) if rg else t
    h = O._bf(x) if ra else x
    def bn(y, b):
        mean = y.mean(dim=(0,2,3)); var = y.var(dim=(0,2,3), unbiased=False)
        scale = sd[f"{b}.weight"] * torch.rsqrt(var + O.BN_EPS)
        if shift:
            # store y - c with c = a per-channel constant close to the batch mean (here: the mean rounded to bf16)
            c = O._bf(mean.detach())[None,:,None,None]
            ys = R(y - c, ry) + c
        else:
            ys = R(y, ry)
        return ys * scale[None,:,None,None] + (sd[f"{b}.bias"] - mean*scale)[None,:,None,None]
    for i in range(1,5):
        for j in (0,3):
            w = sd[f"conv{i}.net.{j}.weight"]; w = rf(w) if rw else w
            y = F.conv2d(h, w, sd[f"conv{i}.net.{j}.bias"], padding=1)
            a = torch.relu(bn(y, f"conv{i}.net.{j+1}"))
            if j == 3 and i <= 3: a = F.max_pool2d(a, 2)
            h = R(a, ra)
    flat = torch.flatten(h, 1)
    wm = sd["fc_mu.weight"]; wl = sd["fc_logvar.weight"]
    if rw: wm, wl = rf(wm), rf(wl)
    mu = F.linear(flat, wm, sd["fc_mu.bias"]); logvar = F.linear(flat, wl, sd["fc_logvar.bias"])
    z = O.reparameterize(mu, logvar, eps)
    if ra: z = rf(z)
    wu = sd["fc_up.weight"]; wu = rf(wu) if rw else wu
    hw = int(round(math.sqrt(wu.shape[0]//256)))
    h = R(torch.tanh(F.linear(z, wu, sd["fc_up.bias"])), ra).view(-1,256,hw,hw)
    for k,(up,b) in enumerate((("up3","bn3"),("up2","bn2"),("up1","bn1"))):
        w = sd[f"{up}.weight"]; w = rf(w) if rw else w
        y = F.conv_transpose2d(h, w, sd[f"{up}.bias"], stride=2)
        a = torch.relu(bn(y, b))
        h = R(a, ra) if k < 2 else a
    return torch.sigmoid(F.conv2d(h, sd["head.weight"], sd["head.bias"])), mu, logvar

def run(x, sd, eps, **kw):
    params = {k: v.detach().clone().requires_grad_(True) for k,v in sd.items() if O.is_param(k)}
    full = dict(sd); full.update(params)
    recon, mu, logvar = fwd(x, full, eps, **kw)
    loss = O.vae_loss(recon, x, mu, logvar, "l1", 1.0)
    gl = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    return recon.detach(), mu.detach(), logvar.detach(), {k:(g if g is not None else torch.zeros_like(params[k])) for k,g in zip(params, gl)}

def rel(a,b): return ((a-b).norm()/b.norm()).item()
torch.manual_seed(1337)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sd = O.construct_state(3,3,128,0.0,32)
sd = O.perturb_norm_params(sd, 7)
x, eps = O.make_inputs(B,3,32,128,1337)
ref = run(x, sd, eps, rw=False, ry=False, ra=False, rg=False)
out = {}
for name, kw in [("all", dict(rw=1,ry=1,ra=1,rg=1)), ("all_shift", dict(rw=1,ry=1,ra=1,rg=1,shift=True)),
                 ("fwd_only(no grad rounding)", dict(rw=1,ry=1,ra=1,rg=0)),
                 ("W only", dict(rw=1,ry=0,ra=0,rg=0)), ("Y only", dict(rw=0,ry=1,ra=0,rg=0)), ("Y only shift", dict(rw=0,ry=1,ra=0,rg=0,shift=True)),
                 ("A only", dict(rw=0,ry=0,ra=1,rg=0)), ("W+A", dict(rw=1,ry=0,ra=1,rg=0)), ("W+A+G", dict(rw=1,ry=0,ra=1,rg=1))]:
    r = run(x, sd, eps, **kw)
    gw = {k: rel(r[3][k], ref[3][k]) for k in ref[3] if ref[3][k].norm() > 1e-4 and not (k.endswith('.bias') and ('net.0' in k or 'net.3' in k or k.startswith('up')))}
    worst = max(gw, key=gw.get)
    allg = rel(torch.cat([r[3][k].flatten() for k in gw]), torch.cat([ref[3][k].flatten() for k in gw]))
    out[name] = dict(recon=rel(r[0],ref[0]), mu=rel(r[1],ref[1]), logvar=rel(r[2],ref[2]), grad_all=allg, grad_worst=gw[worst], worst=worst,
                     c1w=gw.get('conv1.net.0.weight'))
    print(name, json.dumps(out[name]))
os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
json.dump({"note": "relative L2 against the fp32 oracle, batch %d, 32x32, latent 128; rw / ry / ra / rg = bf16 rounding of weights / "
                   "stored pre-BatchNorm tensors / stored activations / stored gradients" % B, "variants": out},
          open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'rounding_ablation.json'), 'w'), indent=1)
