"""The entry point that allocates nothing (include/smrt_hip.h: smrt_vae_generate_noalloc) against the entry point it is a
variant of (smrt_vae_generate): same kernels, same Philox stream, so the outputs must be bit-identical -- on three calls
in a row, the later ones smaller (the static scratch keeps the first call's contents), for a net on the fused inference
path and for nets on the per-layer path (which also take their two activation buffers from the scratch)."""
import numpy as np
import pytest

from oracle.vae_oracle import OracleVAE, flatten_params

pytestmark = pytest.mark.gpu


def _handle(D, hidden, Z, rs):
    from smrt_amd import _hip
    o = OracleVAE(D, hidden, Z, seed=5)
    o.params = [((W + 0.05 * rs.normal(size=W.shape)).astype(np.float32), (b + 0.05 * rs.normal(size=b.shape)).astype(np.float32))
                for W, b in o.params]
    h = _hip.VaeHandle(n_features=D, hidden=hidden, n_latent=Z, activation="sigmoid", optimizer="rms_prop", learning_rate=0.01,
                       l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10, dropout=1.0, seed=5,
                       max_batch=64, device=0)
    h.set_params(flatten_params(o.params))
    return o, h


@pytest.mark.parametrize("D,hidden,Z,n1,n2", [(64, [128], 16, 40000, 777),        # fused path: scratch = z only
                                              (200, [300], 7, 3000, 129),          # per-layer path: z + two activation buffers
                                              (30, [20, 12], 5, 515, 1)])
def test_generate_noalloc_equals_generate(D, hidden, Z, n1, n2):
    import torch
    from smrt_amd import _hip
    rs = np.random.RandomState(D)
    o, h = _handle(D, hidden, Z, rs)
    cap = h.generate_noalloc_max_rows
    assert cap >= n1 > n2
    ld = _hip.rup4(D)
    # offset 2 is not a whole Philox block: also the fused net draws z into the scratch (first, large call); the second call
    # is block-aligned (the fused net draws inside its decode launch, as smrt_vae_generate does, and leaves the scratch
    # alone); the third, smaller and unaligned again, finds the first call's bytes in the scratch
    n3, off3 = max(n2 // 2, 1), 4 * n1 * Z + 6
    for n, off in ((n1, 2), (n2, 4 * n1 * Z + 8), (n3, off3)):
        want = h.generate(n, philox_offset=off, loc=0.25, scale=1.5, to_host=False)
        out = torch.full((n + 3, ld), -7.0, dtype=torch.float32, device=h.device)
        torch.cuda.synchronize()
        h.generate_noalloc(out, n, philox_offset=off, loc=0.25, scale=1.5)
        assert torch.equal(out[:n, :D], want), "no-alloc generate differs from smrt_vae_generate (n = %d)" % n
        assert bool((out[n:] == -7.0).all())                       # rows past n are the caller's
    # and the numbers are the oracle's (the bar of the parity tests)
    from oracle.vae_oracle import STREAM_GENERATE, philox_normal
    z = 0.25 + 1.5 * philox_normal(5, STREAM_GENERATE, 0, off3, n3 * Z).reshape(n3, Z)
    np.testing.assert_allclose(out[:n3, :D].cpu().numpy(), o.decode(z), rtol=1e-4, atol=1e-6)
    with pytest.raises(_hip.HipLibraryError, match="exceed"):
        big = torch.empty((1, ld), dtype=torch.float32, device=h.device)
        h.generate_noalloc(big, cap + 1)
    h.close()
