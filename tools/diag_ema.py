import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_train_iter_gpu import _tiny
from oracle import htrvt_oracle as O
from htrvt_amd.ema import ModelEma
dtype = torch.bfloat16
cfg, sd, m = _tiny(dtype)
x, _, _ = O.synthetic_batch(2, 64, 512, 80, cfg.num_patches, seed=5)
xd = x.cuda()
with torch.no_grad():
    m.eval()
    a, b = m(xd), m(xd)
    print("same model twice equal:", torch.equal(a, b), float((a - b).abs().max()))
    _, _, m2 = _tiny(dtype)
    m2.load_state_dict(m.state_dict()); m2.eval()
    c = m2(xd)
    print("two models same weights equal:", torch.equal(a, c), float((a - c).abs().max()))
    m.train()
    ema = ModelEma(m, 0.9999)
    y0 = ema.ema(xd).clone()
    for k, v in m.state_dict().items():
        if v.dtype != torch.int64 and k != "pos_embed":
            v.add_(torch.randn_like(v) * 0.05 * (v.abs().mean() + 1e-3))
    ema.update(m, num_updates=0)
    y1 = ema.ema(xd).clone()
    _, _, fresh = _tiny(dtype)
    fresh.load_state_dict(ema.ema.state_dict(), strict=True)
    yf = fresh.eval()(xd)
    print("after update vs fresh:", float((y1 - yf).abs().max()))
    ema.ema._engines.clear()
    y2 = ema.ema(xd)
    print("after clearing engines vs fresh:", float((y2 - yf).abs().max()), " vs y1:", float((y2 - y1).abs().max()))
    sa, sb = ema.ema.state_dict(), fresh.state_dict()
    for k in sa:
        if not torch.equal(sa[k], sb[k]):
            print("state differs:", k, sa[k].dtype, float((sa[k].double() - sb[k].double()).abs().max()))
    print("training flags:", ema.ema.training, fresh.training, {type(mm).__name__ for mm in ema.ema.modules() if mm.training}, {type(mm).__name__ for mm in fresh.modules() if mm.training})
    # activations stage by stage through the two engines
    Pa = dict(ema.ema.state_dict(keep_vars=True)); Pb = dict(fresh.state_dict(keep_vars=True))
    ea, eb = ema.ema._engine(xd.device), fresh._engine(xd.device)
    ya = ea.forward(Pa, xd, train=False, save=True); sva = ea.saved
    yb = eb.forward(Pb, xd, train=False, save=True); svb = eb.saved
    print("engine.forward(save=True) diff:", float((ya - yb).abs().max()))
    def cmp(tag, u, v):
        if isinstance(u, torch.Tensor) and isinstance(v, torch.Tensor) and u.shape == v.shape:
            d = float((u.double() - v.double()).abs().max())
            if d > 0: print("  act differs:", tag, d)
    for k in sva:
        if isinstance(sva[k], torch.Tensor): cmp(k, sva[k], svb[k])
    for i, (ba, bb) in enumerate(zip(sva["stem_blocks"], svb["stem_blocks"])):
        for k in ba:
            if isinstance(ba[k], torch.Tensor): cmp(f"blk{i}.{k}", ba[k], bb[k])
            if isinstance(ba[k], tuple):
                for j, (u, v) in enumerate(zip(ba[k], bb[k])): cmp(f"blk{i}.{k}[{j}]", u, v)
    for i, (ba, bb) in enumerate(zip(sva["enc"], svb["enc"])):
        for k in ba:
            if isinstance(ba[k], torch.Tensor): cmp(f"enc{i}.{k}", ba[k], bb[k])
    eng = fresh._engine(xd.device)
    # which stage differs first: compare packed weights
    e1 = ema.ema._engine(xd.device)
    for name in e1._packs:
        t1, t2 = e1._packs[name][1], eng._packs[name][1]
        t1 = t1 if isinstance(t1, tuple) else (t1,)
        t2 = t2 if isinstance(t2, tuple) else (t2,)
        for u, v in zip(t1, t2):
            if not torch.equal(u, v):
                print("pack differs:", name, float((u.float() - v.float()).abs().max()))
