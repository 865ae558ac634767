"""-m gpu, row f3: the device side of the uint8 feed path (rmcl_image_u8_to_patches + the selection derived from the known
extents) against the float path it replaces (collate -> rmcl_patch_select -> rmcl_im2patch_f32 / _sel), and one training step
on a byte batch against the same step on the float batch."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

import rmcl_pkg  # noqa: F401,E402
from oracle import rmcl_oracle as O  # noqa: E402
from rmcl_amd._lib import lib, check, P  # noqa: E402
from rmcl_amd.vilt.datasets import RawUint8Batch, Uint8Batch, select_from_sizes  # noqa: E402
from rmcl_amd.vilt.transforms import normalize_lut  # noqa: E402
from tests.test_parity2_gpu import make_module  # noqa: E402
from tests.test_path_gpu import dev_batch  # noqa: E402

DEV = "cuda:0"


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def byte_batch(sizes, Hmax, Wmax, seed):
    g = torch.Generator().manual_seed(seed)
    data = torch.zeros(len(sizes), Hmax, Wmax, 3, dtype=torch.uint8)
    for b, (h, w) in enumerate(sizes):
        data[b, :h, :w] = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
    return Uint8Batch(data, torch.tensor(sizes, dtype=torch.int32))


@pytest.mark.parametrize("sizes,Hmax,Wmax", [([(384, 384)] * 3, 384, 384), ([(384, 352), (320, 384), (224, 288), (32, 32)], 384, 384),
                                             ([(64, 96), (96, 128)], 96, 128)])
def test_u8_to_patches_equals_float_pipeline(sizes, Hmax, Wmax):
    u8 = byte_batch(sizes, Hmax, Wmax, 3)
    B, gh, gw = len(sizes), Hmax // 32, Wmax // 32
    img = u8.float_image().to(DEV)                                   # = collate of the float samples (tests/test_feed_cpu.py)
    sel_k = torch.empty(B, gh * gw, dtype=torch.int32, device=DEV)
    cnt_k = torch.empty(B, dtype=torch.int32, device=DEV)
    hw_k = torch.empty(B, 2, dtype=torch.int32, device=DEV)
    check(lib.rmcl_patch_select(P(img), B, 3, Hmax, Wmax, 32, P(sel_k), P(cnt_k), P(hw_k), stream()))
    sel, counts, hw = select_from_sizes(u8.sizes, gh, gw)
    assert torch.equal(sel, sel_k.cpu()) and torch.equal(counts, cnt_k.cpu()) and torch.equal(hw, hw_k.cpu())   # extents == pixel mask
    n = int(counts.max())
    ref = torch.empty(B * n, 3072, device=DEV)
    check(lib.rmcl_im2patch_sel(P(img), P(ref), P(sel_k), P(cnt_k), gh * gw, B, n, 3, Hmax, Wmax, 32, 0, stream()))
    out = torch.full((B * n, 3072), 7.0, device=DEV)
    lut, data, sz = normalize_lut().to(DEV), u8.data.to(DEV), u8.sizes.to(DEV)
    check(lib.rmcl_image_u8_to_patches(P(data), P(sz), P(sel_k), P(cnt_k), gh * gw, B, n, Hmax, Wmax, 32, P(lut), P(out), stream()))
    assert torch.equal(out, ref)
    if all(s == (Hmax, Wmax) for s in sizes):                        # dense form: no selection
        dense = torch.empty(B * gh * gw, 3072, device=DEV)
        check(lib.rmcl_im2patch_f32(P(img), P(dense), B, 3, Hmax, Wmax, 32, 0, stream()))
        out2 = torch.empty_like(dense)
        check(lib.rmcl_image_u8_to_patches(P(data), P(sz), None, None, 0, B, gh * gw, Hmax, Wmax, 32, P(lut), P(out2), stream()))
        assert torch.equal(out2, dense)


@pytest.mark.parametrize("sizes", [[(384, 384)] * 4, [(384, 352), (320, 384), (224, 288), (384, 384)]])
def test_training_step_on_a_byte_batch_equals_the_float_batch(sizes):
    ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=len(sizes), adv_steps_img=2)
    batch = O.synthetic_batch(ocfg, len(sizes), 4, ragged_text=True)
    u8 = byte_batch(sizes, 384, 384, 9)
    res = []
    for kind in ("float", "bytes"):
        m, _ = make_module(ocfg, 5, "bf16", k_seed=6)
        b = dev_batch(dict(batch, image=[u8.float_image()]))
        if kind == "bytes":
            b["image"] = [u8]                                        # host bytes: bind_batch moves and normalises them
        m.zero_grad()
        loss = m.training_step(b, 0)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), m.engine.g32.clone(), m.proj_queue.clone()))
    # identical patch rows in, so identical loss / keys; the gradient arena up to the order of its float atomics (bias sums, embeddings)
    # (the batch loss is an atomic sum over the rows: equal up to the order of 4 float adds)
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0]) and torch.equal(res[0][2], res[1][2])
    assert float((res[0][1] - res[1][1]).norm() / res[0][1].norm()) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# MinMaxResize on the device (round 4): rmcl_image_resize_u8 against PIL itself
# ---------------------------------------------------------------------------------------------------------------------

def raw_batch(sizes, seed, shorter=384, longer=640):
    import numpy as np
    rng = np.random.default_rng(seed)
    Hs, Ws = max(h for h, _ in sizes), max(w for _, w in sizes)
    data = torch.zeros(len(sizes), Hs, Ws, 3, dtype=torch.uint8)
    for b, (h, w) in enumerate(sizes):
        yy, xx = np.mgrid[0:h, 0:w]
        smooth = 127 + 90 * np.sin(xx / 17.0)[..., None] * np.cos(yy / 23.0)[..., None]          # edges and gradients: the filter's negative lobes
        img = (smooth + rng.normal(0, 40, (h, w, 3))).clip(0, 255).astype(np.uint8)               # overshoot both ways, the clip to a byte is exercised
        data[b, :h, :w] = torch.from_numpy(img)
    return RawUint8Batch(data, torch.tensor(sizes, dtype=torch.int32), shorter, longer)


@pytest.mark.parametrize("sizes", [[(480, 640), (640, 480), (427, 640), (375, 500)], [(300, 451), (60, 50), (801, 799), (200, 1024), (900, 1200)],
                                   [(384, 384), (384, 512)]])
def test_device_min_max_resize_gives_pils_bytes(sizes):
    """Decoded bytes at their original sizes -> rmcl_image_resize_u8 (two integer passes with PIL's tables) against PIL.Image.resize on
    the same bytes (the reference's MinMaxResize, vilt/transforms/utils.py:5-26): bit-identical inside every sample, zero outside; up-
    and down-scaling, mixed orientations, identity axes, the 640-pixel cap."""
    ocfg = O.default_config(num_layers=1, num_negative=256, per_gpu_batchsize=len(sizes), adv_steps_img=1)
    m, _ = make_module(ocfg, 5, "bf16")
    raw = raw_batch(sizes, 7)
    want = raw.resized_on_host()                                       # PIL
    got = m.engine.resize_raw(raw.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(got.sizes, want.sizes) and tuple(got.data.shape) == tuple(want.data.shape)
    assert torch.equal(got.data.cpu(), want.data)
    if sizes[0] == (300, 451):                                          # and the reference's own pixels of this image (tests/golden/pipeline.npz)
        from tests.golden_util import load
        g = load("pipeline.npz")
        one = RawUint8Batch(torch.from_numpy(g["pix_src"])[None], torch.tensor([[300, 451]]), 384, 640)
        out = m.engine.resize_raw(one.to(DEV))
        t = out.float_image()[0].cpu()
        assert float((t[:, ::16, ::16] - torch.from_numpy(g["pix_out_sub"])).abs().max()) <= 1e-6


def test_training_step_on_decoded_bytes_equals_the_host_resized_batch():
    """batch["image"] = RawUint8Batch (what collate_raw_uint8 delivers: workers only decode) through training_step against the same batch
    resized on the host with PIL and fed as a Uint8Batch: identical patch rows, so identical loss, queue and gradients (up to atomics)."""
    sizes = [(480, 640), (640, 480), (427, 640), (375, 500)]
    ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=len(sizes), adv_steps_img=2)
    batch = O.synthetic_batch(ocfg, len(sizes), 4, ragged_text=True)
    raw = raw_batch(sizes, 11)
    res = []
    for kind in ("host", "device"):
        m, _ = make_module(ocfg, 5, "bf16", k_seed=6)
        b = dev_batch(dict(batch, image=[raw.resized_on_host().float_image()]))
        b["image"] = [raw.resized_on_host() if kind == "host" else raw]
        m.zero_grad()
        # (427 x 640 resizes to 384 x 576 = 216 patches > max_image_len = 200: like the reference, a random subset of 200 is kept, drawn
        # from the HOST generator (vision_transformer.py:633-636) - the same seed gives both runs the same subset)
        torch.manual_seed(123)
        loss = m.training_step(b, 0)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), m.engine.g32.clone(), m.proj_queue.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0]) and torch.equal(res[0][2], res[1][2])
    assert float((res[0][1] - res[1][1]).norm() / res[0][1].norm()) < 1e-6


def test_itm_step_on_decoded_bytes_uses_the_device_resize():
    """The ITM + word-patch-alignment objective mixes "image" and "false_image_0" pixel by pixel (objectives.py:722-730), so both must share
    ONE extent: collate_raw_uint8's batch-wide extent, honoured by the device resize.  Raw batches against the same batches resized on the
    host with PIL: identical ITM / WPA losses."""
    from rmcl_amd.vilt.datasets import RawUint8Batch
    sizes, fsizes = [(480, 640), (640, 480), (427, 640), (375, 500)], [(375, 500), (480, 640), (640, 480), (427, 640)]
    ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=4, adv_steps_img=1)
    batch = O.synthetic_batch(ocfg, 4, 4, ragged_text=True)
    raw, fraw = raw_batch(sizes, 21), raw_batch(fsizes, 22)
    ext = tuple(max(a, b) for a, b in zip(raw.out_hw(), fraw.out_hw()))
    raw, fraw = RawUint8Batch(raw.data, raw.sizes, 384, 640, ext), RawUint8Batch(fraw.data, fraw.sizes, 384, 640, ext)
    labels = torch.tensor([1, 0, 1, 0])
    res = []
    for kind in ("host", "device"):
        m, _ = make_module(ocfg, 5, "bf16", k_seed=6, itm=1)
        m.itm_labels_override = labels
        m.current_tasks = ["itm"]
        b = dev_batch(dict(batch))
        b["image"] = [raw.resized_on_host() if kind == "host" else raw]
        b["false_image_0"] = [fraw.resized_on_host() if kind == "host" else fraw]
        torch.manual_seed(7)                                         # (384 x 576 = 216 patches > max_image_len: the subset is drawn on the host)
        from rmcl_amd.vilt.modules import objectives
        m.engine.dropout_on = False
        m.step_sync.begin_step()
        ret = objectives.compute_itm_wpa(m, b)
        torch.cuda.synchronize()
        res.append((float(ret["itm_loss"]), float(ret["itm_wpa_loss"])))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0]) and abs(res[0][1] - res[1][1]) <= 1e-6 * max(abs(res[0][1]), 1e-3), res
