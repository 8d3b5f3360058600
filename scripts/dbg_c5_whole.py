"""Diagnostic (scratch): whole-block run of one IAO BasicBlock of c5, product vs oracle -- where does g_mid (the gradient at the activation between the two convs) differ?"""
import copy, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import torch_oracle as TO
from micronet_amd.train import build_model, synth_batch
name = sys.argv[1] if len(sys.argv) > 1 else "conv2_x.1"
kw = dict(a_bits=4, w_bits=4, q_type=0, q_level=0)
Q = importlib.import_module("micronet.compression.quantization.wqaq.iao.quantize")
torch.set_num_threads(32)
orc = TO.prepare(build_model("resnet18"), "iao", inplace=True, **kw).train()
pristine = copy.deepcopy(orc)
prod = Q.prepare(build_model("resnet18"), inplace=True, **kw).cuda().train()
get = lambda m, n: eval("m." + n.split(".")[0])[int(n.split(".")[1])]
rec = {}
blk = get(orc, name)
def _blk_hook(mod, i, o):
    rec["in"] = i[0].detach().clone()
    o.register_hook(lambda g: rec.__setitem__("gout", g.detach().clone()))
blk.register_forward_hook(_blk_hook)
x, y = synth_batch(256)
torch.nn.functional.cross_entropy(orc(x), y).backward()
ob, pb = copy.deepcopy(get(pristine, name)).train(), get(prod, name)
cap = {}
def tap(tag, store):
    def fn(mod, i, o):
        store[tag] = o
        if o.requires_grad:
            o.register_hook(lambda g: store.__setitem__("g_" + tag, g.detach().clone()))
    return fn
ob.residual_function[2].register_forward_hook(tap("amid", cap))
xo = rec["in"].clone().requires_grad_(True)
ob(xo).backward(rec["gout"])
capp = {}
pb.residual_function[2].register_forward_hook(tap("amid", capp))
xp = rec["in"].cuda().requires_grad_(True)
pb(xp).backward(rec["gout"].cuda())
ao, ap = cap["amid"].detach(), capp["amid"].detach().cpu()
go, gp = cap["g_amid"], capp["g_amid"].cpu()
print("a_mid: max |diff| / max", float((ao - ap).abs().max() / ao.abs().max()), "max(oracle)", float(ao.max()), "max(product)", float(ap.max()), "bit-equal max:", float(ao.max()) == float(ap.max()))
d = (go - gp).abs()
print("g_mid: max|diff| / max|g|", float(d.max() / go.abs().max()), "elements with diff > 1e-5 max:", int((d > 1e-5 * go.abs().max()).sum()), "of", d.numel())
idx = torch.nonzero(d > 1e-3 * go.abs().max())[:12]
sc_o = float(ob.residual_function[3].aq.scale)
sc_p = float(pb.residual_function[3].activation_quantizer.scale)
print("scale oracle", sc_o, "product", sc_p, "obs max oracle", float(ob.residual_function[3].aq.observer.max_val), "product", float(pb.residual_function[3].activation_quantizer.observer.max_val))
for i in idx:
    i = tuple(int(t) for t in i)
    print(i, "a_o %.9g a_p %.9g  v_o %.7f  g_o %.4e g_p %.4e" % (float(ao[i]), float(ap[i]), float(ao[i]) / sc_o, float(go[i]), float(gp[i])))

print("dx rel", float((xp.grad.cpu() - xo.grad).abs().max() / xo.grad.abs().max()))
pn = dict(pb.named_parameters())
for n_, p_ in ob.named_parameters():
    if p_.grad is not None and n_ in pn and pn[n_].grad is not None:
        print("  d%-40s rel %.3e" % (n_, float((pn[n_].grad.cpu() - p_.grad).abs().max() / p_.grad.abs().max())))
# which g_mid elements differ: relation to the clamp of the NEXT quantizer and to the ReLU
z = ao / sc_o
big = d > 1e-5 * go.abs().max()
print("differing g_mid elements: count", int(big.sum()), " of which a_mid == 0:", int((big & (ao == 0)).sum()), " a/s > 7:", int((big & (z > 7)).sum()), " a/s in (7.4, 7.6):", int((big & (z > 7.4) & (z < 7.6)).sum()))
print("oracle g_mid zero where product nonzero:", int(((go == 0) & (gp != 0)).sum()), " reverse:", int(((go != 0) & (gp == 0)).sum()))
