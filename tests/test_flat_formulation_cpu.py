"""CPU: the product's modules + autograd layer, driven by the torch restatement of the kernel contract,
must reproduce the reference-generated golden fixtures (forward, loss, every gradient, 3 Adam steps,
BatchNorm buffers, eval logits).  This pins the flat/CSR formulation and the hand-derived backward
schedules; the HIP kernels are then checked op by op against the same restatement on the GPU."""
import pytest
import torch

import cgc_net_amd  # noqa: F401
from cgc_net_amd import network
from util import CASES, build_model, load_case, rel_err

TOL = 1e-4        # outputs: the north-star tolerance (fp32, relative)
# Gradients: the reference's OWN fp32 gradients sit up to 8e-5 (relative) away from an fp64 evaluation of the
# same network on the medium fixtures (measured: GCN_embed_3.gcn1.bias), and a second fp32 evaluation order is
# as far on its own -- so two correct fp32 implementations can differ by ~2e-4.  5e-4 is the gradient tolerance.
TOL_GRAD = 5e-4


@pytest.mark.parametrize('name', CASES)
def test_golden_forward_backward(name, torch_kernels):
    cfg, batch, sd, out, grad, sd3 = load_case(name)
    model = build_model(network.SoftPoolingGcnEncoder, cfg, collect_assign=True)
    model.load_state_dict(sd, strict=True)
    model.train()
    logits, loss = model(batch)
    assert rel_err(logits, out['logits']) < TOL
    assert rel_err(loss, out['loss']) < TOL
    for i, s in enumerate(model.assign_matrix):
        assert s.shape == out['assign%d' % (i + 1)].shape
        assert rel_err(s, out['assign%d' % (i + 1)]) < TOL
    loss.backward()
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        assert rel_err(p.grad, grad[k]) < TOL_GRAD, k


@pytest.mark.parametrize('name', CASES)
def test_golden_three_adam_steps(name, torch_kernels):
    cfg, batch, sd, out, grad, sd3 = load_case(name)
    model = build_model(network.SoftPoolingGcnEncoder, cfg)
    model.load_state_dict(sd, strict=True)
    model.train()
    _, loss = model(batch)          # the fixture's recorded sequence: 1 fwd/bwd, then 3 full steps
    loss.backward()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for _ in range(3):
        _, loss = model(batch)
        opt.zero_grad()
        torch.mean(loss).backward()
        opt.step()
    for k, v in model.state_dict().items():
        # Adam's g/sqrt(v) amplifies rounding differences of near-zero gradients: 2e-3 on stepped weights
        tol = 2e-3 if v.dtype.is_floating_point else 0
        if v.dtype.is_floating_point:
            assert rel_err(v, sd3[k]) < tol, k
        else:
            assert int(v) == int(sd3[k]), k
    model.eval()
    with torch.no_grad():
        logits = model(batch)
    assert rel_err(logits, out['eval_logits3']) < 5e-3
