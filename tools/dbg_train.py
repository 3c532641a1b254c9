import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import test_gpu_train as T
sn, g, net, ev, data = T.setup(os.path.join(os.getcwd(), 'tests', 'golden'))
loss, total = T.run_step(g, net, ev, data)
total.backward()
params = dict(net.named_parameters())
for n in ["G_NeRF_net.fc10Col.bias", "G_NeRF_net.fc10Sigma.bias", "G_NeRF_net.fc_sky_color_2.bias", "get_class_layer.bias", "adjust_col.bias", "G_NeRF_net.fc9.norm.bias", "G_NeRF_net.fc10Col.weight"]:
    ref = g["grad_" + n]; got = params[n].grad.cpu().numpy()
    m = np.abs(ref) > 0.2 * np.abs(ref).max()
    print(n, "ratio got/ref:", np.round((got[m] / ref[m]).reshape(-1)[:8], 4))
# ---- independent check of the compositing backward with torch autograd on the engine's own forward tensors
net.zero_grad()
torch.manual_seed(77 + int(g["seed"]))
out = ev.eval(data, net, 0, True)
rho = out["Rho"].detach().clone().requires_grad_(True)
col = out["Col"].detach().clone().requires_grad_(True)
sky = out["Sky_Col"][:, 0, :].detach().clone().requires_grad_(True)
sv = out["Solar_Vis"].detach()
dl = out["deltas"].detach()
y = rho * dl
pv = torch.exp(-(torch.cumsum(y, 1) - y)); pe = 1 - torch.exp(-y); ps = pv * pe
alb = (ps * col).sum(1)
sv3 = torch.sigmoid(((sv * ps).sum(1) - .2) * 30)
rgb = alb * (sv3 + (1 - sv3) * sky)
print("fwd rgb diff", (rgb - out["Rendered_Col"]).abs().max().item())
gt = data["GT_Color"].cuda()
L = torch.mean((rgb - gt) ** 2)
L.backward()
dpre = col.grad * col.detach() * (1 - col.detach())
print("torch-derived fc10Col.bias grad:", dpre.sum((0, 1)).cpu().numpy(), " golden:", g["grad_G_NeRF_net.fc10Col.bias"])
# ---- engine internals after a backward of ONLY the colour loss
import ctypes as C
net.zero_grad()
torch.manual_seed(77 + int(g["seed"]))
out2 = ev.eval(data, net, 0, True)
L2 = torch.mean((out2["Rendered_Col"] - gt) ** 2)
L2.backward()
eng = net._train_engine
def rd(name, n):
    a = np.zeros(n, dtype=np.float32)
    sn._lib.check(eng.L.snerf_trainer_debug_read(eng.h, name.encode(), a.ctypes.data, n), "dbg")
    return a
R, S = 32, 32
e_dcol = rd("d_col", R * S * 3).reshape(R, S, 3); e_drho = rd("d_rho", R * S).reshape(R, S, 1)
print("d_col  engine vs torch: max abs", np.abs(e_dcol - col.grad.cpu().numpy()).max(), " scale", col.grad.abs().max().item())
print("d_rho  engine vs torch: max abs", np.abs(e_drho - rho.grad.cpu().numpy()).max(), " scale", rho.grad.abs().max().item())
print("ratio d_col (first ray)", (e_dcol[0, :4, 0] / col.grad.cpu().numpy()[0, :4, 0]))
e_dhead = rd("d_head", R * S * 4).reshape(R, S, 4)
print("d_head[:, :, 0:3] vs dpre: max abs", np.abs(e_dhead[..., :3] - dpre.cpu().numpy()).max(), "scale", dpre.abs().max().item())
print("colsum d_head:", e_dhead.reshape(-1, 4).sum(0), " engine bias grad:", dict(net.named_parameters())["G_NeRF_net.fc10Col.bias"].grad.cpu().numpy())
