"""tsod_allgather_f32: the RCCL all-gather of the detection records through the C-ABI, on the compute
stream.  One rank on the one GPU of the test box (the gather of a single rank is the identity, but it goes through
ncclCommInitRank + ncclAllGather for real), and two ranks as two processes sharing that GPU is NOT something RCCL allows
(one communicator rank per device), so the world-size-2 control flow stays with the gloo tests of test_dist_gloo.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_allgather_single_rank_through_rccl():
    from two_stage_object_detection_amd.dist import TsodCommunicator
    dev = torch.device("cuda:0")
    comm = TsodCommunicator(rank=0, world=1)
    det = torch.randn(8, 300, 6, device=dev)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):                      # stream-ordered on whatever stream is current
        out = comm.all_gather(det * 2.0)
    side.synchronize()
    assert out.shape == (8, 300, 6) and torch.equal(out, det * 2.0)
    out2 = torch.empty(8, 300, 6, device=dev)
    assert comm.all_gather(det, out=out2) is out2
    torch.cuda.synchronize()
    assert torch.equal(out2, det)
    with pytest.raises(ValueError):
        comm.all_gather(det.half())
    comm.close()
    comm.close()                                       # idempotent
