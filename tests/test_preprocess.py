"""Image transform.  CPU: the numpy restatement of Pillow's 8-bit bicubic resampler is
bit-exact against Pillow itself.  GPU: the HIP transform is bit-exact against the reference's PIL/torch
transform (Resize(224, BICUBIC) -> CenterCrop -> ToTensor -> Normalize), float for float."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle.pil_resize_oracle import resize_bicubic_u8, reference_transform_size

SIZES = [(300, 400), (1279, 1706), (230, 300), (100, 80), (224, 224), (640, 480), (57, 91), (224, 500), (900, 224)]


@pytest.mark.parametrize("h,w", SIZES)
def test_oracle_resize_is_pillow_exact(h, w):
    rng = np.random.RandomState(h * 31 + w)
    img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
    nw, nh = reference_transform_size(w, h, 224)
    ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))
    assert np.array_equal(resize_bicubic_u8(img, nh, nw), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", SIZES)
def test_gpu_transform_bit_exact(h, w):
    from generativeimage2text_amd import engine as E, inference as I
    rng = np.random.RandomState(h * 17 + w)
    img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
    ref = I.image_transform(Image.fromarray(img), 224)                     # the PIL + torch path of the reference
    got = E.preprocess_image(torch.from_numpy(img).cuda(), 224).cpu()
    assert got.shape == ref.shape == (3, 224, 224)
    assert torch.equal(got, ref), (got - ref).abs().max()


@pytest.mark.gpu
def test_gpu_transform_smooth_image_and_repeat():
    # smooth content (real photos are smooth; rounding ties behave differently than on noise) + coefficient cache reuse
    from generativeimage2text_amd import engine as E, inference as I
    yy, xx = np.mgrid[0:777, 0:1031]
    img = np.stack([(yy * 255 // 776), (xx * 255 // 1030), ((yy + xx) % 256)], -1).astype(np.uint8)
    ref = I.image_transform(Image.fromarray(img), 224)
    for _ in range(2):
        got = E.preprocess_image(torch.from_numpy(img).cuda(), 224).cpu()
        assert torch.equal(got, ref)


MINMAX_SIZES = [(480, 640), (1279, 1706), (333, 500), (500, 333), (200, 1000), (700, 525), (97, 61)]


@pytest.mark.parametrize("h,w", MINMAX_SIZES)
def test_oracle_minmax_resize_is_pillow_exact(h, w):
    """the resize of MinMaxResizeForTest (inference.py:29-64) through the numpy restatement of Pillow's resampler"""
    from generativeimage2text_amd.inference import MinMaxResizeForTest
    rng = np.random.RandomState(h * 13 + w)
    img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
    t = MinMaxResizeForTest(480, 640)
    oh, ow = t.get_size((w, h))
    ref = np.asarray(t(Image.fromarray(img)))
    assert ref.shape == (oh, ow, 3)
    assert np.array_equal(resize_bicubic_u8(img, oh, ow), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", MINMAX_SIZES)
@pytest.mark.parametrize("mn,mx", [(480, 640), (420, 560)])
def test_gpu_minmax_transform_bit_exact(h, w, mn, mx):
    from generativeimage2text_amd import inference as I
    rng = np.random.RandomState(h * 19 + w)
    img = Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8))
    ref = I.minmax_image_transform(img, mn, mx)                             # PIL + torch, as the reference does it
    got = I.gpu_minmax_image_transform(img, mn, mx).cpu()
    assert got.shape == ref.shape
    assert torch.equal(got, ref), (got - ref).abs().max()


@pytest.mark.gpu
def test_gpu_batch_transform_bit_exact_with_the_reference_transform():
    """gitmi_preprocess_batch: the decoded images of a batch, of ANY sizes, behind each other in ONE device buffer -> one launch
    pair per 24 images.  Every image must come out float for float as the reference's PIL + torch transform gives it (and
    therefore as the per-image kernels do): landscape, portrait, no-resize, one-pass-only and tiny images, more images than
    one launch chunk holds, unaligned offsets."""
    from generativeimage2text_amd import engine as E, inference as I
    rng = np.random.RandomState(3)
    sizes = SIZES + [(224, 300), (300, 224), (480, 640)] * 6 + [(225, 224), (64, 64)]          # 29 images: two launch chunks
    imgs = [rng.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    offs, total = [], 0
    for im in imgs:
        offs.append(total)
        total += im.size + 7                       # deliberately unaligned
    host = np.zeros(total, dtype=np.uint8)
    for o, im in zip(offs, imgs):
        host[o:o + im.size] = im.reshape(-1)
    out = E.preprocess_batch(torch.from_numpy(host).cuda(), [(o, im.shape[0], im.shape[1]) for o, im in zip(offs, imgs)], 224).cpu()
    assert out.shape == (len(imgs), 3, 224, 224)
    for i, im in enumerate(imgs):
        ref = I.image_transform(Image.fromarray(im), 224)
        assert torch.equal(out[i], ref), (i, im.shape, (out[i] - ref).abs().max())
    # an image that does not lie inside the staging buffer is refused, not read
    with pytest.raises(E.GitmiError, match="does not fit the staging buffer"):
        E.preprocess_batch(torch.from_numpy(host).cuda(), [(total - 10, 64, 64)], 224)
