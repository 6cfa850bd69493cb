"""One tiny masked-pretraining step of the fused HIP ViT on cuda:0, checked against the oracle
(CPU restatement, bf16 autocast).  Checker code: lives under tests/ because it imports the oracle;
used by __graft_entry__.smoke()."""
import torch


def run():
    from mem_amd.modeling_pretrain import pt_vit
    from mem_amd.optim_factory import FlatAdamW, get_parameter_groups
    from oracle.gen_golden import TINY, vit_inputs
    from oracle.vit_ref import RefViT, fill_by_name, make_optimizer, train_step
    import contextlib, io
    m = pt_vit(**TINY)
    w = fill_by_name(m.state_dict(), seed=0)
    m.load_state_dict(w)
    m = m.cuda().train()
    o = RefViT(**TINY)
    o.load_state_dict(w)
    x, mask, labels = vit_inputs(TINY, 4, 3, 6)
    with contextlib.redirect_stdout(io.StringIO()):
        opt = FlatAdamW(m, get_parameter_groups(m, 0.05, m.no_weight_decay()), lr=5e-4)
    oopt = make_optimizer(o)
    ref_loss, ref_norm, _ = train_step(o, oopt, x, mask, labels, 0, clip_grad=30.0, autocast_dtype=torch.bfloat16)
    la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda())
    m.backward()
    gn = m.engine.grad_norm()
    opt.max_norm = 30.0
    opt.step()
    assert abs(la[0].item() - ref_loss) <= 3e-3, (la[0].item(), ref_loss)
    assert abs(gn.item() / ref_norm - 1) <= 0.03, (gn.item(), ref_norm)
    for (k, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
        d = (p.detach().cpu() - q.detach()).abs().max().item()
        assert d <= 2e-3, (k, d)      # lr 5e-4 Adam first step: |dp| <= lr; sign flips on ~0 grads allowed
    print(f"model smoke ok: loss {la[0].item():.4f} (oracle {ref_loss:.4f}), |g| {gn.item():.4f} (oracle {ref_norm:.4f})")
