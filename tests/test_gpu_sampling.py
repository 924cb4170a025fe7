"""Incremental (KV-cache) decoding == the reference's full-pass-per-pixel schedule, position by position,
and the end-to-end sampler contract."""
import pytest
import torch

import seeded
from conftest import rel_err
from oracle import lvt_oracle as O
from util_models import dsfvt_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vt():
    from lvt_amd.modeling import build_model
    cfg = dsfvt_cfg()
    cfg.TEST.EVALUATORS = "VTSampler"
    model = build_model(cfg)
    model.model.load_state_dict(seeded.seeded_params(seeded.dsfvt_shapes(), 4321), strict=False)
    model.eval()
    return model


def test_incremental_step_equals_full_pass_rows(vt, golden):
    from lvt_amd.modeling.autoregressive.incremental import IncrementalDecoder
    codes = torch.stack([seeded.seeded_codes("s%d" % i, (16, 4, 16, 16), 3) for i in range(3)])
    data = [O.prepare_slices(codes[i], (6, 0, 0), (16, 1, 1), (7, 1, 1), 1) for i in range(3)]
    ctx = torch.stack([d["context"] for d in data]).to(DEV)
    sl = torch.stack([d["slice"] for d in data]).to(DEV)
    si = torch.stack([d["slice_idx"] for d in data]).to(DEV)
    with torch.no_grad():
        zl = vt.model.encoder.forward_tokens(ctx, si)
        full = vt.model.decoder.forward_tokens(sl, zl).view(3, 256, 512)
        dec = IncrementalDecoder(vt.model.decoder, zl, 3, (1, 16, 16))
        worst = 0.0
        for i in range(256):
            y = dec.step(sl, i)
            if i in (0, 1, 15, 16, 17, 100, 254, 255):
                worst = max(worst, rel_err(y, full[:, i]))
        assert worst < 2e-5, worst
        # teacher-forced probabilities of the golden pixel set through the incremental path
        g = golden("g13_sample_probs")
        g12 = golden("g12_dsfvt_loss")
        d = O.prepare_slices(g12["codes"][0], (3, 0, 0), (16, 1, 1), (7, 1, 1), 1)
        ctx1, sl1, si1 = d["context"][None].to(DEV), d["slice"][None].to(DEV), d["slice_idx"][None].to(DEV)
        zl1 = vt.model.encoder.forward_tokens(ctx1, si1)
        dec1 = IncrementalDecoder(vt.model.decoder, zl1, 1, (1, 16, 16))
        for i in range(256):
            y = dec1.step(sl1, i)
            hi, wi = divmod(i, 16)
            if (hi, wi) in ((0, 0), (7, 9), (15, 15)):
                _, probs = vt.model.ch_predictor.sample_from_rows(y, 1.0, forced_codes=sl1[:, :, 0, hi, wi], return_probs=True)
                assert rel_err(probs[0], g["probs_%d_%d" % (hi, wi)]) < 1e-4


def test_sample_videos_contract_and_priming(vt):
    codes = torch.stack([seeded.seeded_codes("v%d" % i, (16, 4, 16, 16), 8) for i in range(2)])
    n_prime = 14                                           # generate the last two frames only (512 positions)
    with torch.no_grad():
        video = codes.transpose(1, 2).contiguous().to(DEV)
        video[:, :, n_prime:] = 0
        torch.manual_seed(0)
        out = vt.sample_video(video, n_prime=n_prime)
        assert tuple(out.shape) == (2, 4, 16, 16, 16) and out.dtype == torch.int64
        assert torch.equal(out[:, :, :n_prime].cpu(), codes.transpose(1, 2)[:, :, :n_prime])      # primed frames untouched
        assert 0 <= int(out.min()) and int(out.max()) < 512
        assert int((out[:, :, n_prime:] != 0).sum()) > 0.9 * out[:, :, n_prime:].numel()
        # the reference schedule draws from the same distributions: with the same seed and temperature -> 0 both
        # paths pick the arg-max codes, which must coincide wherever the top-2 probability gap is not a rounding tie
        a = vt.sample_video(video, n_prime=15, temp=1e-4, incremental=True)
        b = vt.sample_video(video, n_prime=15, temp=1e-4, incremental=False)
        assert float((a != b).float().mean()) < 0.01
    # the inference-mode contract used by generate_videos.py / VTSampler
    vt.cfg.TEST.VT_SAMPLER.N_PRIME = 15
    with torch.no_grad():
        res = vt([{"image_sequence": codes[0]}], mode="inference")
    assert len(res) == 1 and len(res[0]["samples"]) == 1 and tuple(res[0]["samples"][0].shape) == (4, 16, 16, 16)


def test_large_batches_decode_as_stream_groups(vt):
    """More than 64 videos are decoded as independent groups of <= 64 on separate streams; every group must produce
    exactly what it produces alone (arg-max sampling, so the draws do not depend on the random stream)."""
    import lvt_amd.modeling.meta_arch.vt as vtmod
    B = 70                                                   # two groups of 35
    monkey_rows = vtmod.DECODE_GROUP_ROWS
    vtmod.DECODE_GROUP_ROWS = 64
    try:
        video, both = _check_stream_groups(vt, vtmod, B)
    finally:
        vtmod.DECODE_GROUP_ROWS = monkey_rows
        vt._samplers = {}
    # the same 70 videos as ONE group: every decode launch then covers two 64-row blocks
    with torch.no_grad():
        one = vt.sample_video(video, n_prime=15, temp=1e-4)
    assert len(vt._samplers[(B, 1, 16, 16, 1e-4)]) == 1
    assert float((one != both)[:, :, 15:].float().mean()) < 0.01


def _check_stream_groups(vt, vtmod, B):
    codes = torch.stack([seeded.seeded_codes("g%d" % (i % 5), (16, 4, 16, 16), 8 + i % 7) for i in range(B)])
    with torch.no_grad():
        video = codes.transpose(1, 2).contiguous().to(DEV)
        video[:, :, 15:] = 0
        vt._samplers = {}
        torch.manual_seed(11)
        both = vt.sample_video(video, n_prime=15, temp=1e-4)
        assert len(vt._samplers[(B, 1, 16, 16, 1e-4)]) == 2
        lo = vt.sample_video(video[:35].contiguous(), n_prime=15, temp=1e-4)
        hi = vt.sample_video(video[35:].contiguous(), n_prime=15, temp=1e-4)
    assert torch.equal(both[:, :, :15].cpu(), codes.transpose(1, 2)[:, :, :15])
    # concurrent groups == the same groups run one after the other on one stream, bit for bit
    vtmod.DECODE_GROUP_STREAMS = False
    try:
        vt._samplers = {}
        torch.manual_seed(11)                 # same uniforms: near-ties of the arg-max are decided by them
        with torch.no_grad():
            serial = vt.sample_video(video, n_prime=15, temp=1e-4)
    finally:
        vtmod.DECODE_GROUP_STREAMS = True
        vt._samplers = {}
    print("serial vs concurrent groups: %d codes differ" % int((serial != both).sum()))
    assert torch.equal(serial, both)
    # waves (more groups than may run concurrently): one group per wave here; the uniforms are consumed in another
    # order, so only arg-max near-ties may differ
    vtmod.MAX_CONCURRENT_GROUPS = 1
    try:
        vt._samplers = {}
        with torch.no_grad():
            waves = vt.sample_video(video, n_prime=15, temp=1e-4)
    finally:
        vtmod.MAX_CONCURRENT_GROUPS = 3
        vt._samplers = {}
    assert float((waves != both)[:, :, 15:].float().mean()) < 0.01
    # the encoder pass over 70 vs 35 videos may pick another split-K count (last-bit differences in the context), so
    # arg-max near-ties of this random-init model can flip; anything beyond that would be a race between the groups
    mism = float((torch.cat([lo, hi]) != both)[:, :, 15:].float().mean())
    print("grouped vs alone: %.4f of the generated codes differ" % mism)
    assert mism < 0.01
    return video, both


def test_sample_categorical_matches_oracle_rule():
    """lvt_sample_categorical == oracle.multinomial_from_uniform on softmax(logits / temp) (rows whose threshold
    is not within rounding of a cdf step), and its probabilities == softmax."""
    from lvt_amd.hip import tx
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(300, 512, generator=g) * 3
    u = torch.rand(300, generator=g)
    u[0], u[1] = 0.0, 0.999999
    temp = 0.9
    prob = torch.softmax(logits.double() / temp, 1)
    want = O.multinomial_from_uniform(prob, u.double())
    cdf = torch.cumsum(prob, 1)
    margin = (cdf - (u.double() * cdf[:, -1]).unsqueeze(1)).abs().min(1).values
    out = torch.full((300, 4), -1, dtype=torch.int64, device=DEV)
    pr = tx.sample_categorical(logits.to(DEV), temp, u.to(DEV), out.view(-1)[2:], 4, want_probs=True)
    got = out[:, 2].cpu()
    safe = margin > 1e-5
    assert int(safe.sum()) > 280
    assert torch.equal(got[safe], want[safe])
    assert bool(((got - want).abs() <= 1).all())
    assert bool((out[:, [0, 1, 3]] == -1).all())
    assert rel_err(pr, prob.float()) < 1e-5


def test_graph_replay_equals_eager_steps(vt, monkeypatch):
    """One captured hipGraph per group replayed for every position (device-side cursor) == the same steps launched
    eagerly, bit for bit: two generated frames (512 positions, the second frame replays the graph from position 0)."""
    codes = torch.stack([seeded.seeded_codes("e%d" % i, (16, 4, 16, 16), 21) for i in range(3)])
    with torch.no_grad():
        video = codes.transpose(1, 2).contiguous().to(DEV)
        video[:, :, 14:] = 0
        vt._samplers = {}
        torch.manual_seed(5)
        graphed = vt.sample_video(video, n_prime=14, temp=1e-4)
        (_, _, smp, _), = vt._samplers[(3, 1, 16, 16, 1e-4)]
        assert set(smp.graphs) == {True} and smp._next == 256          # ONE graph served 2 x 256 positions
        monkeypatch.setenv("LVT_DECODE_GRAPHS", "0")
        vt._samplers = {}
        torch.manual_seed(5)
        eager = vt.sample_video(video, n_prime=14, temp=1e-4)
        (_, _, smp, _), = vt._samplers[(3, 1, 16, 16, 1e-4)]
        assert not smp.graphs
        vt._samplers = {}
    print("graph replay vs eager: %d codes differ" % int((graphed != eager).sum()))
    assert torch.equal(graphed, eager)


def test_workspace_refuses_capture():
    """The grow-and-replace scratch of binding.workspace() must never be recorded into a graph."""
    from lvt_amd.hip import binding as L
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with pytest.raises(L.LvtError, match="capture"):
        with torch.cuda.graph(g):
            L.workspace(1 << 20, torch.device(DEV), "gemm")


@pytest.mark.parametrize("ngroups", [1, 3])
def test_groups_of_256_videos(vt, ngroups):
    """The bench configuration of the generation leg (groups of 256 videos, three groups replaying their graphs
    concurrently on three streams) for one generated frame: concurrent == the groups one after the other on one stream,
    bit for bit; every code in range; primed frames untouched."""
    import lvt_amd.modeling.meta_arch.vt as vtmod
    B = 256 * ngroups
    base = torch.stack([seeded.seeded_codes("w%d" % i, (16, 4, 16, 16), 30 + i) for i in range(16)])
    codes = base[torch.arange(B) % 16].roll(1, 0).contiguous()
    with torch.no_grad():
        video = codes.transpose(1, 2).contiguous().to(DEV)
        video[:, :, 15:] = 0
        vt._samplers = {}
        torch.manual_seed(3)
        conc = vt.sample_video(video, n_prime=15, temp=1e-4)
        groups = vt._samplers[(B, 1, 16, 16, 1e-4)]
        assert len(groups) == ngroups and all(set(g[2].graphs) == {True} for g in groups)
        assert torch.equal(conc[:, :, :15].cpu(), codes.transpose(1, 2)[:, :, :15])
        assert 0 <= int(conc.min()) and int(conc.max()) < 512
        # identical inputs 16 videos apart inside a group: same arithmetic per row, only arg-max near-ties (decided by the
        # per-row uniforms) may differ
        assert float((conc[0:16] != conc[16:32]).float().mean()) < 0.01
        vtmod.DECODE_GROUP_STREAMS = False
        try:
            vt._samplers = {}
            torch.manual_seed(3)
            serial = vt.sample_video(video, n_prime=15, temp=1e-4)
        finally:
            vtmod.DECODE_GROUP_STREAMS = True
            vt._samplers = {}
    print("%d groups of 256: serial vs concurrent %d codes differ" % (ngroups, int((serial != conc).sum())))
    assert torch.equal(serial, conc)
