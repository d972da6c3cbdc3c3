"""TEST INFRASTRUCTURE ONLY (checker for ppst_amd/smooth_filter.py; never imported by the product path).

CPU restatement (numpy) of the reference's local-affine photo smoothing, /root/reference/smooth_filter.py:
  best_local_affine_kernel   :149-241   per pixel, least-squares 3x4 affine map  content patch -> stylised patch
  bilateral_smooth_kernel    :243-290   joint-bilateral average of the 12 affine coefficients (guide = content)
  reconstruction_best_kernel :293-321   apply the smoothed map to the content pixel
  smooth_local_affine        :332-378   driver (sigma1 = f_r / 3, sigma2 = f_e, radius = (patch - 1) / 2)
  smooth_filter              :381-405   PIL front end (BGR flip, /255, clip * 255 -> uint8 truncation)

PARITY UNPINNED: the reference runs these kernels through cupy + pynvrtc (CUDA only; neither package nor a CUDA device
exists here), the function is dead code in the reference (nothing calls it), and the reference holds no test or fixture
for it.  This file follows the kernel source's arithmetic statement by statement -- float32 products accumulated in
float64 where the kernel adds a float product to a double, float32 `exp`, float32 storage of both affine tables -- but
CUDA's expf and nvcc's fused multiply-add contraction are not reproducible bit for bit on a CPU, so the comparison bar
against the HIP kernels is a tolerance (tests/test_gpu_parity.py: 1e-6 absolute on [0, 1] images), not equality.
"""

import numpy as np

F32 = np.float32

def _shifted(a, dy, dx):
    """a[..., y+dy, x+dx] with a validity mask (False where the neighbour falls outside the image)."""
    H, W = a.shape[-2:]
    out = np.zeros_like(a)
    ys, ye = max(0, -dy), min(H, H - dy)
    xs, xe = max(0, -dx), min(W, W - dx)
    valid = np.zeros((H, W), dtype=bool)
    if ys < ye and xs < xe:
        out[..., ys:ye, xs:xe] = a[..., ys + dy:ye + dy, xs + dx:xe + dx]
        valid[ys:ye, xs:xe] = True
    return out, valid

def _inverse4x4(m):
    """adjugate / determinant in float64 (InverseMat4x4, smooth_filter.py:10-147): returns (inverse, ok) with
    ok = |det| >= 1e-9; failed pixels get an all-zero inverse (the kernel leaves invMt_M at its zero initialisation)."""
    n = m.shape[0]
    adj = np.zeros_like(m)
    idx = [0, 1, 2, 3]
    for i in range(4):
        for j in range(4):
            rows = [r for r in idx if r != j]
            cols = [c for c in idx if c != i]
            sub = m[:, rows][:, :, cols]
            d3 = (sub[:, 0, 0] * (sub[:, 1, 1] * sub[:, 2, 2] - sub[:, 1, 2] * sub[:, 2, 1])
                  - sub[:, 0, 1] * (sub[:, 1, 0] * sub[:, 2, 2] - sub[:, 1, 2] * sub[:, 2, 0])
                  + sub[:, 0, 2] * (sub[:, 1, 0] * sub[:, 2, 1] - sub[:, 1, 1] * sub[:, 2, 0]))
            adj[:, i, j] = d3 * (1.0 if (i + j) % 2 == 0 else -1.0)
    det = (m[:, 0, :] * adj[:, :, 0]).sum(-1)
    ok = np.abs(det) >= 1e-9
    inv = np.zeros_like(m)
    inv[ok] = adj[ok] / det[ok][:, None, None]
    assert inv.shape == (n, 4, 4)
    return inv, ok

def best_local_affine(output, input_, radius=1):
    """(3,H,W) float32 stylised `output`, content `input_` -> affine_model (H*W, 12) float32 (smooth_filter.py:149-241).
    Feature vector f = (I[2], I[1], I[0], 1); row i of the model predicts output channel 2 - i."""
    output, input_ = output.astype(F32), input_.astype(F32)
    _, H, W = input_.shape
    f = [input_[2], input_[1], input_[0], None]
    t = [output[2], output[1], output[0]]
    MtM = np.zeros((H, W, 4, 4), np.float64)
    MtS = np.zeros((H, W, 3, 4), np.float64)
    for i in range(3):
        MtM[..., i, i] = 1e-3
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            fs = []
            valid = None
            for c in range(3):
                v, valid = _shifted(f[c], dy, dx)
                fs.append(v)
            ts = [_shifted(tc, dy, dx)[0] for tc in t]
            one = valid.astype(np.float64)
            for a in range(4):
                for b in range(4):
                    if a < 3 and b < 3:
                        MtM[..., a, b] += np.where(valid, (fs[a] * fs[b]).astype(F32), F32(0)).astype(np.float64)
                    elif a < 3:
                        MtM[..., a, 3] += np.where(valid, fs[a], F32(0)).astype(np.float64)
                    elif b < 3:
                        MtM[..., 3, b] += np.where(valid, fs[b], F32(0)).astype(np.float64)
                    else:
                        MtM[..., 3, 3] += one
            for i in range(3):
                for j in range(3):
                    MtS[..., i, j] += np.where(valid, (fs[j] * ts[i]).astype(F32), F32(0)).astype(np.float64)
                MtS[..., i, 3] += np.where(valid, ts[i], F32(0)).astype(np.float64)
    inv, _ = _inverse4x4(MtM.reshape(-1, 4, 4))
    S = MtS.reshape(-1, 3, 4)
    # A[i][j] = sum_k inv[j][k] * Mt_S[i][k]
    A = np.einsum("njk,nik->nij", inv, S)
    return A.reshape(-1, 12).astype(F32)

def bilateral_smooth(affine_model, guide, radius, sigma1, sigma2):
    """(H*W,12) float32 model, (3,H,W) float32 guide -> filtered model (H*W,12) float32 (smooth_filter.py:243-290)."""
    guide = guide.astype(F32)
    _, H, W = guide.shape
    am = affine_model.reshape(H, W, 12).transpose(2, 0, 1).astype(F32)
    sum_a = np.zeros((12, H, W), np.float64)
    sum_w = np.zeros((H, W), np.float64)
    s1, s2 = F32(sigma1), F32(sigma2)
    den1, den2 = F32(2) * s1 * s1, F32(2) * s2 * s2
    for dx in range(-radius, radius + 1):
        for dy in range(-radius, radius + 1):
            g, valid = _shifted(guide, dy, dx)
            if not valid.any():
                continue
            d = g - guide
            cds = ((d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / F32(3)).astype(F32)
            v1 = np.exp(F32(-(dx * dx + dy * dy)) / den1).astype(F32)
            v2 = np.exp(-cds / den2).astype(F32)
            wgt = np.where(valid, (v1 * v2).astype(F32), F32(0))
            a, _ = _shifted(am, dy, dx)
            sum_a += (wgt[None] * a).astype(F32).astype(np.float64)
            sum_w += wgt.astype(np.float64)
    out = (sum_a / sum_w[None]).astype(F32)
    return out.transpose(1, 2, 0).reshape(-1, 12)

def reconstruction(input_, filtered_model):
    """(3,H,W) content, (H*W,12) model -> (3,H,W) float32: channel c = row c of the model applied to (I[2], I[1], I[0], 1)
    (smooth_filter.py:293-321; float32 expression, contracted left to right as nvcc's default -fmad does)."""
    input_ = input_.astype(F32)
    _, H, W = input_.shape
    m = filtered_model.reshape(H, W, 12).astype(F32)
    i2, i1, i0 = (input_[2].astype(np.float64), input_[1].astype(np.float64), input_[0].astype(np.float64))
    out = np.zeros((3, H, W), F32)
    for c in range(3):
        a = [m[..., 4 * c + k].astype(np.float64) for k in range(4)]
        p = (i2 * a[0]).astype(F32).astype(np.float64)
        p = (i1 * a[1] + p).astype(F32).astype(np.float64)      # fma
        p = (i0 * a[2] + p).astype(F32).astype(np.float64)      # fma
        out[c] = (p + a[3]).astype(F32)
    return out

def smooth_local_affine(output, input_, epsilon, patch, h, w, f_r, f_e):
    """smooth_filter.py:332-378 (epsilon is passed to the kernel and unused there)."""
    assert output.shape == (3, h, w) and input_.shape == (3, h, w)
    radius = int((patch - 1) / 2)
    model = best_local_affine(output, input_, radius)
    filt = bilateral_smooth(model, input_, int(f_r), f_r / 3, f_e)
    return reconstruction(input_, filt)

def smooth_filter_arrays(init_rgb_u8, content_rgb_u8, f_radius=15, f_edge=1e-1):
    """smooth_filter.py:381-405 on uint8 (H,W,3) RGB arrays of equal size -> uint8 (H,W,3) RGB."""
    best = np.ascontiguousarray(init_rgb_u8[:, :, ::-1].transpose(2, 0, 1).astype(F32)) / F32(255.)
    cont = np.ascontiguousarray(content_rgb_u8[:, :, ::-1].transpose(2, 0, 1).astype(F32)) / F32(255.)
    _, H, W = cont.shape
    r = smooth_local_affine(best, cont, 1e-7, 3, H, W, f_radius, f_edge).transpose(1, 2, 0)
    return np.uint8(np.clip(r * F32(255.), 0, 255.))
