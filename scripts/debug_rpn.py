import sys, torch
sys.path.insert(0, '/root/repo')
import oracle
from two_stage_object_detection_amd import hip_ops
from two_stage_object_detection_amd.testing import synthetic_detector
dev = torch.device('cuda:0')
bb = sys.argv[1] if len(sys.argv) > 1 else 'hardnet39'
model, sd = synthetic_detector(bb, num_classes=20, seed=0)
model = model.to(dev).eval()
x = torch.rand((2, 3, 320, 448), generator=torch.Generator().manual_seed(1234))
with torch.inference_mode():
    ref_out, dbg = oracle.detector_forward(sd, x, backbone=bb, return_debug=True)
    feat = dbg['feat']
    rpn = model.rpn
    f = hip_ops.nchw_to_nhwc(feat.to(dev))
    n, h, w, _ = f.shape
    pc_loc, pc_score, base = rpn._pack(dev)
    locs = hip_ops.conv2d_nhwc(f, pc_loc.w, shift=pc_loc.shift)
    scores = hip_ops.conv2d_nhwc(f, pc_score.w, shift=pc_score.shift)
    A = base.shape[0]
    boxes, fg, keys, anchor = hip_ops.rpn_decode(locs.view(n*h*w, 4*A), scores.view(n*h*w, 2*A), base, n, h, w, rpn.feat_stride, 320, 448, 16.0, True)
    counts, idx, bs, ks = hip_ops.sort_topk_desc(keys, boxes, 3000)
    keep, rois, n_kept, status = hip_ops.nms_sorted(bs, counts, 0.7, 300)
    for b in range(n):
        d = dbg['per_image'][b]
        print('img', b, 'fg maxdiff', float((fg[b].cpu()-dbg['fg'][b]).abs().max()),
              'box maxdiff', float((boxes[b].cpu()-d['decoded']).abs().max()),
              'valid mismatch', int((torch.isfinite(keys[b].cpu()) != d['valid']).sum()),
              'count', int(counts[b]), len(d['sorted_src']))
        c = int(counts[b])
        gi = idx[b,:c].cpu().long(); ri = d['sorted_src']
        neq = (gi != ri).nonzero().flatten()
        print('  sorted idx mismatches', neq.numel(), neq[:10].tolist())
        for j in neq[:6].tolist():
            print('    pos', j, 'gpu', int(gi[j]), 'ref', int(ri[j]), 'score gpu', float(fg[b].cpu()[gi[j]]), float(fg[b].cpu()[ri[j]]), 'ref scores', float(dbg['fg'][b][gi[j]]), float(dbg['fg'][b][ri[j]]))
        # NMS on the oracle's sorted boxes through GPU
        k2, r2, nk2, st2 = hip_ops.nms_sorted(d['roi_sorted'].unsqueeze(0).to(dev), torch.tensor([c], dtype=torch.int32, device=dev), 0.7, 300)
        kref = d['keep']
        print('  nms(oracle sorted boxes) keep equal', torch.equal(k2[0].cpu().long(), kref), 'n_kept', int(nk2[0]), d['n_kept'])
        kg = keep[b].cpu().long()
        neqk = (kg != kref).nonzero().flatten()
        print('  pipeline keep mismatches', neqk.numel(), neqk[:10].tolist(), 'first vals', kg[neqk[:5]].tolist(), kref[neqk[:5]].tolist())
        print('  rois rows mismatched', int(((rois[b].cpu()-ref_out[2][b]).abs().amax(-1) > 1e-3).sum()))
