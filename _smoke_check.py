"""One tiny pass of the hot path on the GPU, checked against the CPU oracle (used by
__graft_entry__.smoke()).  The oracle is only the checker here."""
import torch


def run(device):
    from oracle import dswin_oracle as D
    from visiontransformer_amd.dswin import DeformableSparseWindowAttention, WindowedMSA

    torch.manual_seed(0)
    w, C, heads = 8, 24, 2
    x = torch.randn(2, 32, 32, C)

    def rel(a, b):
        return ((a.detach().cpu().double() - b.detach().double()).norm() / b.detach().double().norm()).item()

    for cls in (WindowedMSA, DeformableSparseWindowAttention):
        mod = cls(w, C, heads)
        with torch.no_grad():
            for p in mod.parameters():
                p.normal_(0, 0.2)
            if cls is DeformableSparseWindowAttention:
                mod.sparse_discriminator[4].weight.mul_(4.0)
                mod.sparse_discriminator[4].bias[1] += 0.5
                mod.num_win_selected = 0.5
        P = {k: v.detach().clone() for k, v in mod.state_dict().items()}
        xc = x.clone().requires_grad_()
        if cls is WindowedMSA:
            ref = D.wmsa_forward(P, xc, w, heads)
        else:
            aux = {}
            ref = D.dsmsa_forward(P, xc, w, heads, training=False, num_win_selected=0.5, aux=aux)
        ref.sum().backward()
        mod = mod.to(device).eval()
        xg = x.to(device).requires_grad_()
        y = mod(xg)
        y.sum().backward()
        torch.cuda.synchronize()
        e_y, e_g = rel(y, ref), rel(xg.grad, xc.grad)
        if cls is DeformableSparseWindowAttention:
            sel = mod.last_selection
            n = sel.n if sel.n_dev is None else int(sel.n_dev.item())
            assert sel is not None and torch.equal(sel.win_h1.cpu().long()[:n], aux["h1"]) \
                and torch.equal(sel.win_w1.cpu().long()[:n], aux["w1"]), "window selection differs from the oracle"
        assert e_y < 1e-3 and e_g < 1e-3, "%s: rel err fwd %.2e bwd %.2e" % (cls.__name__, e_y, e_g)
        print("smoke %-34s fwd rel err %.2e  bwd rel err %.2e" % (cls.__name__, e_y, e_g))
