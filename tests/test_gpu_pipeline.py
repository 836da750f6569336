"""Input side on the GPU (scope row f-3): resize + pad + flips + normalise + CHW in one launch, through the C ABI, against
the vectors captured from the reference's dataset classes (tests/golden/pipeline*.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def lib():
    return np.load(os.path.join(GOLD, 'pipeline.npz'))


@pytest.fixture(scope='module')
def demo():
    return np.load(os.path.join(GOLD, 'pipeline_demo.npz'))


def _ann(z, i):
    return [tuple(r) for r in z[f'p_{i}_ann'].tolist()]


def _sample(ds, rgb, ann, hf, vf):
    label = ds.labels_for(ann, rgb.shape[:2], hf, vf)
    out = torch.zeros([len(label), 6], dtype=torch.float32)
    out[:, 1:] = torch.from_numpy(label)
    return rgb, out, (hf, vf)


def test_library_samples_match_reference(lib):
    from fastvision_amd.datasets import BaseDataset
    for i, (h, w, S, n) in enumerate(lib['p_cases']):
        ds = BaseDataset([], int(S), 200)
        combos = [(hf, vf) for hf in (0, 1) for vf in (0, 1)]
        batch = [_sample(ds, lib[f'p_{i}_rgb'], _ann(lib, i), bool(hf), bool(vf)) for hf, vf in combos]
        images, labels = ds.collate_fn(batch, DEV)                      # the four flip variants as one batch
        images, labels = images.cpu().numpy(), labels.cpu().numpy()
        for k, (hf, vf) in enumerate(combos):
            ref_lab = lib[f'l1_{i}_{hf}{vf}_lab']
            got = labels[labels[:, 0] == k]
            np.testing.assert_array_equal(got[:, 1:], ref_lab[:, 1:])
            if S <= 128:
                np.testing.assert_array_equal(images[k], lib[f'l1_{i}_{hf}{vf}_img'])
            else:
                np.testing.assert_array_equal(images[k][:, ::7, ::5], lib[f'l1_{i}_{hf}{vf}_img_sub'])
                sums = np.array([images[k].astype(np.float64).sum(), (images[k].astype(np.float64) ** 2).sum()])
                np.testing.assert_allclose(sums, lib[f'l1_{i}_{hf}{vf}_img_sums'], rtol=1e-12)


def test_library_collate_matches_reference(lib):
    from fastvision_amd.datasets import BaseDataset
    ds = BaseDataset([], 128, 200)
    batch = [_sample(ds, lib[f'p_{i}_rgb'], _ann(lib, i), False, False) for i in lib['l2_index']]
    images, labels = ds.collate_fn(batch, DEV)
    np.testing.assert_array_equal(labels.cpu().numpy(), lib['l2_labels'])
    np.testing.assert_allclose(images.double().sum(dim=(1, 2, 3)).cpu().numpy(), lib['l2_image_sums'], rtol=1e-12)


def test_dataloader_from_files(tmp_path, lib):
    """create_dataloader end to end on PNG files (lossless) + label files, against the oracle"""
    from PIL import Image
    from fastvision_amd.datasets import create_dataloader
    from oracle import pipeline as P
    os.makedirs(tmp_path / 'images')
    os.makedirs(tmp_path / 'labels')
    idx = [0, 1, 2, 5]
    for i in idx:
        Image.fromarray(lib[f'p_{i}_rgb']).save(tmp_path / 'images' / f'im{i}.png')
        with open(tmp_path / 'labels' / f'im{i}.txt', 'w') as f:
            for r in lib[f'p_{i}_ann']:
                f.write(' '.join(repr(float(v)) for v in r) + '\n')
    loader = create_dataloader('train', str(tmp_path), batch_size=4, input_size=128, device=torch.device(DEV), num_workers=0,
                               cache=None, shuffle=False)
    loader.dataset.flip_p = (-1.0, -1.0)
    batches = list(loader)
    assert len(batches) == 1
    images, labels = batches[0]
    assert images.shape == (4, 3, 128, 128) and images.is_cuda and labels.is_cuda
    ref = P.collate([P.library_sample(lib[f'p_{i}_rgb'], _ann(lib, i), 128, 0, 0) for i in idx])
    np.testing.assert_array_equal(images.cpu().numpy(), ref[0])
    np.testing.assert_array_equal(labels.cpu().numpy(), ref[1])


def test_demo_val_and_mosaic_match_reference(demo):
    from fastvision_amd.demos.yolov3_u.data_gen import DeviceAugmenter
    from fastvision_amd.pipeline_ops import value_table
    table = value_table(single=True)[0]
    samples = []
    for i in range(4):
        ann = demo[f'q_{i}_ann']
        samples.append((demo[f'q_{i}_rgb'], ann[:, 1:].copy(), ann[:, 0].copy()))
    # validation path of one image with both flips applied first = fixture d1 (ResizeByMax -> flips -> Padding)
    aug = DeviceAugmenter(96, DEV)
    from fastvision_amd.pipeline_ops import PasteJob, pack_images, paste_batch
    from fastvision_amd.demos.yolov3_u.data_gen import resize_by_max_shape
    for i, (rgb, xyxy, cat) in enumerate(samples):
        buf, off, shp = pack_images([rgb])
        ratio, rh, rw = resize_by_max_shape(rgb.shape[0], rgb.shape[1], 96)
        job = PasteJob(0, 0, rh, rw, (96 - rh) // 2, (96 - rw) // 2, True, True)
        img = paste_batch(buf, off, shp, [job], 1, 96, 96, 128, aug.table, DEV)[0].cpu().numpy()
        np.testing.assert_array_equal(img, table[demo[f'd1_{i}_padded'].transpose(2, 0, 1)])
    images, labels = aug.val_batch(samples)
    assert images.shape == (4, 3, 96, 96) and labels.shape[1] == 6
    # Mosaic01
    for case, S in enumerate((96, 160)):
        aug = DeviceAugmenter(S, DEV)
        group = [(rgb, xyxy.copy(), cat.copy(), False, False) for rgb, xyxy, cat in samples]
        images, labels = aug.train_batch([group, group[::-1]])
        assert images.shape == (2, 3, S, S)
        np.testing.assert_array_equal(images[0].cpu().numpy(), table[demo[f'd2_{case}_image'].transpose(2, 0, 1)])
        lab0 = labels[labels[:, 0] == 0].cpu().numpy()
        from oracle.pipeline import xyxy2xywhn
        np.testing.assert_array_equal(lab0[:, 2:], xyxy2xywhn(demo[f'd2_{case}_xyxy'], S, S).astype(np.float32))
        np.testing.assert_array_equal(lab0[:, 1], demo[f'd2_{case}_cat'])


def test_full_size_batch_against_oracle():
    """B = 32 at 640x640 from mixed-size sources: equal to the oracle everywhere; reports the kernel's rate"""
    from fastvision_amd.datasets import BaseDataset
    from oracle import pipeline as P
    from oracle.make_golden import synth_image, synth_boxes
    g = np.random.default_rng(1)
    sizes = [(480, 640), (427, 640), (640, 480), (375, 500), (1280, 1280), (640, 640), (333, 500), (720, 1280)] * 4
    ds = BaseDataset([], 640, 200)
    batch, refs = [], []
    for k, (h, w) in enumerate(sizes):
        rgb, ann = synth_image(g, h, w), synth_boxes(g, 3, h, w)
        hf, vf = bool(k & 1), bool(k & 2)
        batch.append(_sample(ds, rgb, ann, hf, vf))
        if k < 8:
            refs.append(P.library_sample(rgb, ann, 640, hf, vf))
    host = ds.collate_host(batch)
    images, labels = ds.to_device(host, DEV)
    for k, (img, lab) in enumerate(refs):
        np.testing.assert_array_equal(images[k].cpu().numpy(), img)
        np.testing.assert_array_equal(labels[labels[:, 0] == k][:, 1:].cpu().numpy(), lab[:, 1:])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    src = host[0].to(DEV)
    host_dev = (src,) + host[1:]
    e0.record()
    for _ in range(10):
        ds.to_device(host_dev, DEV)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nbytes = src.numel() + images.numel() * 4
    print(f'\npaste_kernel B=32 640x640: {ms * 1e3:.0f} us per batch, {nbytes / ms / 1e6:.0f} GB/s (source bytes once + fp32 batch written)')


def test_faster_rcnn_batch_matches_oracle_pipeline():
    """demos/faster_rcnn/data_gen.py: ResizeByMax -> HorizontalFlip (per sample) -> Padding(128) -> / 255, against the CPU
    restatement of those steps (oracle/pipeline.py: the classes of the two demos' data_gen.py are the same code)."""
    from fastvision_amd.demos.faster_rcnn.data_gen import DeviceAugmenter
    from fastvision_amd.demos.yolov3_u.utils.box import xyxy2xywhn
    from oracle import pipeline as P
    rng = np.random.default_rng(3)
    S = 160
    samples, want_img, want_lab = [], [], []
    for i, (h, w) in enumerate([(97, 140), (200, 120), (160, 160), (33, 310)]):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        x1, y1 = rng.uniform(0, w * 0.6, 3), rng.uniform(0, h * 0.6, 3)
        xyxy = np.stack([x1, y1, x1 + rng.uniform(2, w * 0.4, 3), y1 + rng.uniform(2, h * 0.4, 3)], 1).astype(np.float32)
        cat = rng.integers(0, 20, 3).astype(np.float32)
        hflip = bool(i % 2)
        samples.append((rgb, xyxy, cat, hflip))
        img, lab = P.demo_resize_by_max(rgb, xyxy.copy(), S)
        if hflip:
            img, lab = P.demo_hflip(img, lab)
        img, lab = P.demo_padding(img, lab.astype(np.float32), S, 128)
        want_img.append(img.transpose(2, 0, 1).astype(np.float32) / 255.)
        want_lab.append(np.concatenate([np.full((3, 1), i, np.float32), cat[:, None], xyxy2xywhn(lab, S, S).astype(np.float32)], 1))
    images, labels = DeviceAugmenter(S, device=DEV).batch(samples)
    assert tuple(images.shape) == (4, 3, S, S) and tuple(labels.shape) == (12, 6)
    np.testing.assert_allclose(images.cpu().numpy(), np.stack(want_img), rtol=0, atol=1e-7)
    np.testing.assert_allclose(labels.cpu().numpy(), np.concatenate(want_lab), rtol=1e-6, atol=1e-6)
