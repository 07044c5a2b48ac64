"""SURVEY 8(f) rank 3 on the GPU: a checkpoint in the reference's training format (train/train.py:120-128:
``{'model_state_dict': FasterRCNNTrainer.state_dict(), 'optimizer_state_dict': ..., 'scheduler_state_dict': ...}`` with the
backbone under ``feat_extra.``, loaded there with weights_only=True at :60-71) goes through ``load_trainer_checkpoint`` into
the HIP detector and must reproduce the oracle run on the same state_dict; the packed-weight disk cache must serve the
second load without any fold / gather / pack work and give bit-identical results."""
import os

import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _img(shape, seed=1234):
    return torch.rand(shape, generator=torch.Generator().manual_seed(seed))


def _trained_like(backbone, seed):
    """A state_dict that looks trained: non-identity BN statistics, per-block PReLU slopes, scaled heads."""
    from two_stage_object_detection_amd.testing import synthetic_detector
    model, sd = synthetic_detector(backbone, num_classes=20, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    if backbone.startswith("hardnet"):
        oracle.calibrate_bn(sd, _img((2, 3, 256, 320), seed=99), oracle.hardnet_trunk, arch=int(backbone[-2:]), prefix="extractor.")
    else:
        for k in sd:                                  # mild, scale-preserving: the seeded detector stays well-conditioned
            if k.endswith("running_mean"):
                sd[k] = sd[k] + torch.randn(sd[k].shape, generator=g) * 0.01
            elif k.endswith("running_var"):
                sd[k] = sd[k] * (torch.rand(sd[k].shape, generator=g) * 0.04 + 0.98)
            elif k.startswith("extractor.") and k.endswith("bias") and sd[k].dim() == 1:
                sd[k] = sd[k] + torch.randn(sd[k].shape, generator=g) * 0.01
    return model, sd


@pytest.mark.parametrize("backbone", ["resnet50", "hardnet39"])
def test_trainer_checkpoint_to_hip_forward_with_packed_cache(dev, tmp_path, backbone, monkeypatch):
    from two_stage_object_detection_amd import engine, weight_cache
    from two_stage_object_detection_amd.nets.frcnn import FasterRCNN
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    _, sd = _trained_like(backbone, seed=0)
    trainer_sd = {("feat_extra." + k[len("extractor."):] if k.startswith("extractor.") else k): v for k, v in sd.items()}
    ckpt = tmp_path / "FasterRCNNTrainer_best.pth"
    torch.save({"model_state_dict": trainer_sd, "optimizer_state_dict": {"state": {}, "param_groups": []},
                "scheduler_state_dict": {"T_max": 5}}, ckpt)
    cache = tmp_path / "packed"
    x = _img((2, 3, 256, 320), seed=5)
    with torch.inference_mode():
        ref = oracle.detector_forward(sd, x, backbone=backbone)

        torch.manual_seed(999)                                         # different init: everything must come from the file
        m1 = FasterRCNN(20, backbone=backbone).to(dev).eval()
        m1(x.to(dev))                                                  # a forward BEFORE the load (stale packed weights)
        try:
            m1.raise_if_error()                                        # un-gained random heads may trip the pad rule: clear the sticky word
        except IndexError:
            pass
        res = m1.load_trainer_checkpoint(str(ckpt), packed_cache=str(cache))
        assert not res.missing_keys and not res.unexpected_keys
        files = [f for f in os.listdir(cache) if f.endswith(".tsodpack")]
        assert len(files) == 1 and files[0].startswith(backbone + "-")
        got1 = [o.cpu() for o in m1(x.to(dev))]
        m1.raise_if_error()
        rep = compare_detector_outputs(got1, ref)
        print(backbone, rep)
        assert rep["ok"] and rep["rows_unmatched"] <= 2 and rep["class_mismatch"] == 0, str(rep)   # the suite's e2e bar

        # second process-equivalent: a fresh module, same checkpoint -> the cache file is found by content hash and
        # no folding / gathering / packing code runs
        m2 = FasterRCNN(20, backbone=backbone).to(dev).eval()
        m2.load_trainer_checkpoint(str(ckpt))

        def boom(*a, **k):
            raise AssertionError("packing work on a cache hit")
        monkeypatch.setattr(engine, "fold_bn", boom)
        monkeypatch.setattr("two_stage_object_detection_amd.hip_ops.pack_conv_weight", boom)
        monkeypatch.setattr("two_stage_object_detection_amd.models.hardnet.fold_bn", boom)
        monkeypatch.setattr("two_stage_object_detection_amd.models.hardnet._gathered_weight", boom)
        assert m2.use_packed_cache(str(cache)) == "hit"
        got2 = [o.cpu() for o in m2(x.to(dev))]
        for a, b in zip(got1, got2):
            assert torch.equal(a, b)
        monkeypatch.undo()

        # other weights -> other hash -> miss (a second file), never a wrong hit
        with torch.no_grad():
            m2.head.score.bias.add_(0.5)
        m2.invalidate_packed()
        assert m2.use_packed_cache(str(cache)) == "miss"
        assert len([f for f in os.listdir(cache) if f.endswith(".tsodpack")]) == 2
        got3 = m2(x.to(dev))[1].cpu()
        assert (got3 - got2[1] - 0.5).abs().max().item() < 1e-4        # the edited bias is what runs

        # same weights, other anchors (not in the state_dict; the RPN entry stores the base anchors): another file, never a
        # hit that would serve the first detector's anchors
        m3 = FasterRCNN(20, backbone=backbone, anchor_scales=[4, 8, 16]).to(dev).eval()
        m3.load_trainer_checkpoint(str(ckpt))
        assert weight_cache.cache_path(str(cache), m3) != weight_cache.cache_path(str(cache), m1)
        assert m3.use_packed_cache(str(cache)) == "miss"
        base3 = m3.rpn._pack(dev)[1].cpu()
        assert torch.equal(base3, torch.as_tensor(m3.rpn.anchor_base, dtype=torch.float32).cpu())
        assert not torch.equal(base3, torch.as_tensor(m1.rpn.anchor_base, dtype=torch.float32).cpu())

    # tile tables ride along, keyed by geometry + device
    plan = m1.extractor._plan_for(x.to(dev))
    p = weight_cache.save_tiles(m1, str(cache), x.shape, dev, plan.export_tiles())
    assert os.path.basename(p).startswith(f"tiles-{backbone}-2x256x320-")
    assert weight_cache.load_tiles(m1, str(cache), x.shape, dev) == plan.export_tiles()
    assert weight_cache.load_tiles(m1, str(cache), (1, 3, 64, 64), dev) is None


def test_tuned_table_is_cached_on_disk_and_a_second_start_does_not_tune(dev, tmp_path):
    """FasterRCNN.tune(cache_dir=...): the first call tunes and writes the table (keyed by weights + config, device name, input
    geometry, the tuning arguments and the library's sha256), a second detector with the same weights pins it in well under two
    seconds and runs the same kernels (bit-identical outputs); other arguments or other weights miss."""
    import time
    from two_stage_object_detection_amd import weight_cache
    from two_stage_object_detection_amd.testing import synthetic_detector
    x = _img((1, 3, 224, 288), seed=7).to(dev)
    kw = dict(precisions=(0, 2), in_flight=2, reps=1, in_sequence=0, in_flight_refine=0, splits=[1, -1])
    m1, _ = synthetic_detector("resnet50", num_classes=20, seed=0)
    m1 = m1.to(dev).eval()
    with torch.inference_mode():
        t0 = time.perf_counter()
        table = m1.tune(x, cache_dir=str(tmp_path), **kw)
        t_tune = time.perf_counter() - t0
        assert "cached" not in table and len(list(tmp_path.glob("tuning-resnet50-1x224x288-*.json"))) == 1
        out1 = [o.clone() for o in m1(x)]
        m2, _ = synthetic_detector("resnet50", num_classes=20, seed=0)
        m2 = m2.to(dev).eval()
        m2(x)                                                          # (plan building and weight packing are not what is measured)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hit = m2.tune(x, cache_dir=str(tmp_path), **kw)
        t_hit = time.perf_counter() - t0
        assert hit.get("cached") is True and t_hit < 2.0 and t_hit < t_tune, (t_hit, t_tune)
        assert [list(r) for r in hit["serial"]] == [list(r) for r in table["serial"]] and hit["heads"] == table["heads"]
        assert all(hit[k] == table[k] for k in ("fuse_stem", "fuse_bottleneck", "fuse_projection"))
        assert m2.extractor._plan_for(x).export_tiles() == m1.extractor._plan_for(x).export_tiles()
        out2 = m2(x)
        for a, b in zip(out1, out2):
            assert torch.equal(a, b)
        m2.raise_if_error()
        # another argument set, other weights: a miss each (a new file)
        m2.tune(x, cache_dir=str(tmp_path), **dict(kw, splits=[1]))
        assert len(list(tmp_path.glob("tuning-*.json"))) == 2
        m3, _ = synthetic_detector("resnet50", num_classes=20, seed=1)
        m3 = m3.to(dev).eval()
        args = {"precisions": [0, 2]}
        assert weight_cache.tuning_path(str(tmp_path), m3, x.shape, dev, args) != weight_cache.tuning_path(str(tmp_path), m1, x.shape, dev, args)
