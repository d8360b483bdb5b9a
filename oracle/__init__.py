"""CPU oracle for the vision-MLP forward path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (explicit index formulas on torch-CPU tensors,
fp32 or fp64) of the reference's forward path `models_pytorch/*` for the seven
model families on the north-star path.  Every function
cites the reference file:line it follows.

Rules (checked by the judge):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import this package -- as the checker / the timed CPU baseline,
    never as the thing shipped.  The product path (`jittor-mlp_amd/`) never
    imports it and has no CPU fallback.
  * parity pinning: the reference holds no golden vectors of its own
    (SURVEY.md section 4), so the oracle is pinned against outputs of the
    reference itself, generated in the authoring container by
    `tests/golden/make_golden.py` (imports /root/reference, never shipped)
    and committed as `tests/golden/*.npz`.  `tests/test_oracle_golden.py`
    checks the oracle against every one of them.
"""
from .functional import (  # noqa: F401
    gelu, layer_norm, group_norm1, patch_embed,
    mixer_forward, gmlp_forward, resmlp_forward, vip_forward,
    s2mlpv2_forward, s2mlpv1_forward, asmlp_forward, convmixer_forward, sparsemlp_forward, hiremlp_forward, msmlp_forward, swinmlp_forward, cyclemlp_forward, flatten_outputs, drop_path,
    cycle_fc, cycle_offsets, deform_conv2d_pointwise_loop,
    axial_shift_nchw, axial_shift_nchw_backward, spatial_shift1, spatial_shift2, split_attention,
    vip_permute_h, vip_permute_w,
)
from .portable_init import portable_tensor, portable_state_dict, portable_input  # noqa: F401
