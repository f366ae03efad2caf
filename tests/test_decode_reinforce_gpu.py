"""Decode kernels (softmax-xent, argmax, sampling) and the REINFORCE estimator against the oracle."""
import numpy as np
import pytest

from oracle import convlstm as oc
from oracle import raster as oracle_raster
from helpers_convlstm import make_params

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"


@pytest.mark.parametrize("rows,cols", [(1, 784), (7, 785), (640, 1024), (33, 5), (2048, 785)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_xent_forward_backward_and_argmax(rows, cols, dtype):
    from imagesegmentation_amd.polyrnn import decode
    rs = np.random.RandomState(rows + cols)
    z = rs.randn(rows, cols) * 3
    z[0, :] = 0.0                      # all ties: first index must win
    if cols > 3 and rows > 1:
        z[-1, 2] = z[-1, cols - 1] = 50.0  # two equal maxima: the smaller index wins
    labels = rs.randint(0, cols, [rows])
    w = rs.uniform(0, 2, [rows])
    tz = torch.tensor(z, dtype=dtype, device=DEV).requires_grad_(True)
    zr = tz.detach().double().cpu().numpy()  # what the kernel actually sees (bf16-rounded if bf16)
    loss_rows = decode.softmax_xent_rows(tz, torch.tensor(labels, device=DEV), torch.tensor(w, device=DEV))
    ref_rows = oc.softmax_xent(zr, labels) * w
    assert np.allclose(loss_rows.detach().cpu().numpy(), ref_rows, rtol=2e-5, atol=2e-5)
    loss_rows.mean().backward()
    ref_grad = oc.softmax_xent_backward(zr, labels, w / rows)
    tol = 1e-6 if dtype == torch.float32 else 1e-2 * np.abs(ref_grad).max()
    assert np.abs(tz.grad.double().cpu().numpy() - ref_grad).max() <= max(tol, 1e-6)
    am = decode.argmax_rows(tz.detach())
    assert np.array_equal(am.cpu().numpy(), np.argmax(zr, axis=1))
    assert am[0].item() == 0
    pm = decode.prediction_max(tz.detach(), 28)
    assert np.array_equal(pm.cpu().numpy(), oc.prediction_max(zr, 28))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_masked_logits_stay_finite(dtype):
    """-inf entries (masked classes) take no part: loss, gradient and argmax stay finite wherever a lane's first
    column is -inf, and a row of nothing but -inf reports its first column instead of an out-of-range index."""
    from imagesegmentation_amd.polyrnn import decode
    rs = np.random.RandomState(5)
    rows, cols = 6, 200
    z = rs.randn(rows, cols) * 2
    z[0, :64] = -np.inf          # every lane's first column masked
    z[1, ::2] = -np.inf
    z[2, :] = -np.inf
    z[2, 137] = 1.5              # a single live class
    z[3, :] = -np.inf            # nothing live at all
    labels = np.array([100, 1, 137, 0, 5, 7])
    tz = torch.tensor(z, dtype=dtype, device=DEV).requires_grad_(True)
    zr = tz.detach().double().cpu().numpy()
    loss_rows = decode.softmax_xent_rows(tz, torch.tensor(labels, device=DEV), None)
    got = loss_rows.detach().cpu().numpy()
    live = [0, 1, 2, 4, 5]
    with np.errstate(divide="ignore", invalid="ignore"):
        m = zr[live].max(axis=1, keepdims=True)
        ref = (m[:, 0] + np.log(np.exp(zr[live] - m).sum(axis=1))) - zr[live, labels[live]]
    assert np.all(np.isfinite(got[live])) and np.allclose(got[live], ref, rtol=2e-5, atol=2e-5)
    assert abs(got[2]) < 1e-6    # the only live class is the label: loss 0
    am = decode.argmax_rows(tz.detach()).cpu().numpy()
    assert np.array_equal(am[live], np.argmax(zr[live], axis=1)) and am[3] == 0
    a = decode.sample_rows(tz.detach(), seed=9).cpu().numpy()
    assert a[2] == 137 and 0 <= a[3] < cols and np.all(np.isfinite(zr[live, a[live]]))
    loss_rows[live].sum().backward()
    g = tz.grad.double().cpu().numpy()
    assert np.all(np.isfinite(g[live])) and np.all(g[0, :64] == 0) and np.all(g[1, ::2] == 0)


def test_sampling_follows_softmax():
    from imagesegmentation_amd.polyrnn import decode
    rs = np.random.RandomState(0)
    cols, rows = 11, 200000
    z = rs.randn(cols) * 1.5
    p = np.exp(z - z.max())
    p /= p.sum()
    tz = torch.tensor(np.tile(z, (rows, 1)), dtype=torch.float32, device=DEV)
    a = decode.sample_rows(tz, seed=123).cpu().numpy()
    freq = np.bincount(a, minlength=cols) / rows
    assert np.abs(freq - p).max() < 5 * np.sqrt(p.max() / rows) + 1e-3
    b = decode.sample_rows(tz, seed=123).cpu().numpy()
    c = decode.sample_rows(tz, seed=124).cpu().numpy()
    assert np.array_equal(a, b) and not np.array_equal(a, c)   # counter-based: reproducible per seed
    d = decode.sample_rows(tz[:1000], seed=123, row_offset=5).cpu().numpy()
    assert np.array_equal(d[:-5], a[5:1000])                     # row_offset shifts the stream


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sample_actions_is_sample_rows_as_xy(dtype):
    """isg_sample_actions: the same draw as isg_sample_rows, as (class // P, class % P); the device-side row
    offset adds to the host one; the end-token class P*P comes out as (P, 0)."""
    from imagesegmentation_amd.polyrnn import decode
    P, rows = 7, 3000
    tz = (torch.randn([rows, P * P + 1], device=DEV) * 2).to(dtype)
    tz[::5, P * P] = 30.0  # every fifth row all but surely draws the end token
    a = decode.sample_rows(tz, seed=9, row_offset=77).cpu().numpy()
    xy = decode.sample_actions(tz, 9, P, row_offset=77)
    assert xy.dtype == torch.int32 and tuple(xy.shape) == (rows, 2)
    assert np.array_equal(xy.cpu().numpy(), np.stack([a // P, a % P], axis=1))
    assert np.array_equal(xy[::5].cpu().numpy(), np.tile([[P, 0]], (rows // 5, 1)))
    off = torch.tensor([70], dtype=torch.int64, device=DEV)
    xy2 = decode.sample_actions(tz, 9, P, row_offset=7, row_offset_dev=off)
    assert torch.equal(xy, xy2)
    off += 1
    assert not torch.equal(xy, decode.sample_actions(tz, 9, P, row_offset=7, row_offset_dev=off))


def _load(model, params):
    with torch.no_grad():
        for (w, b), p in zip(model.encoder, params["encoder"]):
            w.copy_(torch.tensor(p["w"]))
            b.copy_(torch.tensor(p["b"]))
        for cell, p in zip(model.rnn.cells, params["rnn"]):
            cell.input_to_state_kernel.copy_(torch.tensor(p["w_x"]))
            cell.hidden_to_state_kernel.copy_(torch.tensor(p["w_h"]))
            cell.bias.copy_(torch.tensor(p["bias"]))
        model.dense_kernel.copy_(torch.tensor(params["dense"]["w"]))
        model.dense_bias.copy_(torch.tensor(params["dense"]["b"]))


def _flat(grads):
    out = []
    for g in grads["encoder"]:
        out += [g["w"], g["b"]]
    for g in grads["rnn"]:
        out += [g["w_h"], g["w_x"], g["bias"]]
    return np.concatenate([a.reshape(-1) for a in out + [grads["dense"]["w"], grads["dense"]["b"]]])


@pytest.mark.parametrize("end_token", [False, True])
def test_policy_update_accumulates_the_reference_gradient(end_token):
    """update(): loss = mean_[B,T](xent * td), raw gradients summed over episodes, Adam on the sum."""
    from imagesegmentation_amd.polyrl.reinforce import ExperimentPolicyEstimator, discount_rewards
    P, T = 28, 10
    rs = np.random.RandomState(3)
    pol = ExperimentPolicyEstimator(P, P, max_timesteps=T, end_token=end_token, device=DEV)
    params = make_params(rs, P, 2, units=pol.num_classes)
    _load(pol, params)
    total = 0
    for episode in range(2):
        n = 3 + episode
        poly = rs.randint(0, P, [n + 2, 2])
        image = np.stack([oracle_raster.create_shape_mask(poly, P)] * 3, axis=2)
        hist = np.stack([oracle_raster.create_history(poly, i + 1, 2, P) for i in range(n)])
        actions = poly[2:2 + n].copy()
        if end_token:
            actions[-1] = (P, 0)
        td = discount_rewards([0.0] * (n - 1) + [0.7])
        loss = pol.update(image, hist, actions, td, apply_grads=False)
        H = np.zeros([1, T, P, P, 2])
        H[0, :n] = hist
        A = np.zeros([1, T, 2], np.int64)
        A[0, :n] = actions
        W = np.zeros([1, T])
        W[0, :n] = td
        ref_loss, _, ref_grads = oc.model_loss_and_grads(params, image[None], H, [n], A, P, weights=W)
        assert abs(loss - ref_loss) < 1e-5 * max(1, abs(ref_loss))
        total = total + _flat(ref_grads)
    got = pol.gradBuffer.double().cpu().numpy()
    assert np.abs(got - total).max() < 2e-4 * np.abs(total).max()
    before = [p.detach().clone() for p in pol.trainable_variables]
    pol.apply_accumulated()
    assert float(pol.gradBuffer.abs().max()) == 0.0 and pol.global_step == 1
    moved = max(float((a - b).abs().max()) for a, b in zip(pol.trainable_variables, before))
    assert 0 < moved <= 1.0001e-4   # one Adam step of lr 1e-4, unclipped


@pytest.mark.parametrize("end_token", [False, True])
def test_batched_rollouts_match_scalar_semantics(end_token):
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl.reinforce import BatchedReinforce, ExperimentPolicyEstimator, discount_rewards
    P, T, N = 28, 10, 64
    ds = data.PolygonDataset.synthetic(32, P, P, T, 2, seed=2, shape_mask_fn=oracle_raster.create_shape_mask)
    pol = ExperimentPolicyEstimator(P, P, max_timesteps=T, end_token=end_token, device=DEV, sample_seed=7)
    with torch.no_grad():
        pol.dense_bias[-1 if end_token else 0] += 4.0   # make terminations frequent enough to test
    br = BatchedReinforce(pol, ds, N)
    np.random.seed(0)
    images, polys = ds.raw_sample(32)
    images = np.concatenate([images, images]); polys = np.concatenate([polys, polys])
    if not end_token:  # closing needs the first true vertex: put it on the class the bias favours, (0, 0)
        for b in range(len(polys)):
            polys[b] = polys[b].copy()
            polys[b][0] = (0, 0)
    ep = br.rollout(images=images, polygons=list(polys))
    actions, lengths = ep["actions"].cpu().numpy(), ep["lengths"].cpu().numpy()
    reward, done = ep["reward"].cpu().numpy(), ep["done"].cpu().numpy()
    points = ep["points"].cpu().numpy()
    td = br.returns(ep).cpu().numpy()
    n_done = 0
    for b in range(N):
        v = polys[b]
        pred = [list(p) for p in v[:2]]
        hist = [list(p) for p in v[:2]]
        finished = False
        for t in range(lengths[b]):
            assert points[b, t].tolist() == hist
            a = actions[b, t].tolist()
            if end_token and a == [P, 0]:
                pred.append(list(v[0])); finished = True
            else:
                pred.append(a); hist = [hist[1], a]   # np.roll(+1) then overwrite the last frame, hist = 2
                if not end_token and a == list(v[0]):
                    finished = True
            if finished:
                assert t == lengths[b] - 1
        assert finished == bool(done[b])
        want = oracle_raster.calculate_iou(np.array(pred), v, P) if finished else 0.0
        assert reward[b] == want
        rewards = [0.0] * max(lengths[b] - 1, 0) + ([want] if lengths[b] else [])
        ref_td = np.zeros([T]); ref_td[:lengths[b]] = discount_rewards(rewards)
        assert np.allclose(td[b], ref_td, rtol=1e-6, atol=1e-7)
        n_done += finished
    assert n_done > 0
    # identical (image, polygon) pairs still explore different actions: the sampler is keyed per row
    assert not np.array_equal(actions[:32], actions[32:])
    loss = br.update(ep)
    assert np.isfinite(float(loss)) and pol.global_step == 1


def test_device_pool_draws_like_raw_sample():
    """DevicePolygonPool.sample: distinct indices while n <= len(dataset), every polygon a roll of its dataset
    polygon (np.roll by a start vertex, reference Dataset.py:158-160), every start reached, images in step."""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl.batched_env import DevicePolygonPool
    P = 28
    ds = data.PolygonDataset.synthetic(24, P, P, 10, 2, seed=4, shape_mask_fn=oracle_raster.create_shape_mask)
    pool = DevicePolygonPool(ds, DEV, vcap=13, seed=3)
    img, vt, cnt, idx = pool.sample(24)
    assert sorted(idx.cpu().tolist()) == list(range(24))
    img2, vt2, cnt2, idx2 = pool.sample(4096)
    seen_starts = set()
    vt2_h, cnt2_h, idx2_h = vt2.cpu().numpy(), cnt2.cpu().numpy(), idx2.cpu().numpy()
    assert np.array_equal(img2[:64].cpu().numpy(), ds._images[idx2_h[:64]].astype(np.float32))
    for b in range(4096):
        p = np.asarray(ds._vertices[idx2_h[b]])
        n = len(p)
        assert cnt2_h[b] == n and not vt2_h[b, n:].any()
        starts = [s for s in range(n) if np.array_equal(np.roll(p, s, axis=0), vt2_h[b, :n])]
        assert starts
        seen_starts.add((int(idx2_h[b]), starts[0]))
    assert len(seen_starts) > 0.9 * sum(len(v) for v in ds._vertices)
    again = DevicePolygonPool(ds, DEV, vcap=13, seed=3)
    assert torch.equal(again.sample(24)[1], vt)


@pytest.mark.parametrize("end_token", [False, True])
def test_rollout_from_device_pool_matches_host_reset(end_token):
    """A rollout whose episodes were drawn on the device equals the rollout handed the same images and polygons
    from the host: same actions, lengths and rewards."""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl.reinforce import BatchedReinforce, ExperimentPolicyEstimator
    P, T, N = 28, 10, 96
    ds = data.PolygonDataset.synthetic(32, P, P, T, 2, seed=5, shape_mask_fn=oracle_raster.create_shape_mask)
    eps = []
    for device_pool in (True, False):
        pol = ExperimentPolicyEstimator(P, P, max_timesteps=T, end_token=end_token, device=DEV, sample_seed=11)
        with torch.no_grad():
            pol.dense_bias[-1 if end_token else 0] += 4.0
        br = BatchedReinforce(pol, ds, N, device_pool=device_pool, pool_seed=8)
        if device_pool:
            ep = br.rollout()
            nv = br.env.num_vertices.cpu().numpy()
            polys = [br.env.vertices[b, :nv[b]].cpu().numpy() for b in range(N)]
            images = ep["images"].cpu().numpy()
        else:
            ep = br.rollout(images=images, polygons=polys)
        eps.append(ep)
    for key in ("actions", "lengths", "reward", "inter", "union", "done", "points"):
        assert torch.equal(eps[0][key], eps[1][key]), key
    assert int(eps[0]["done"].sum()) > 0


def test_scalar_reinforce_loop_schedule():
    """reinforce() (reference polyrl/ReinforceOverfitExperiment.py:279-331) on the drop-in scalar environment: an
    episode is learnt from only when its last step earned a reward, with discounted returns as weights, and
    the gradient buffer is applied on every fifth episode but the first."""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl.polygon_env_end_token import PolygonEnv
    from imagesegmentation_amd.polyrl.reinforce import ExperimentPolicyEstimator, discount_rewards, reinforce
    P, T = 28, 10
    ds = data.PolygonDataset.synthetic(4, P, P, T, 2, seed=6, shape_mask_fn=oracle_raster.create_shape_mask)
    pol = ExperimentPolicyEstimator(P, P, max_timesteps=T, end_token=True, device=DEV, sample_seed=3)
    with torch.no_grad():
        pol.dense_bias[-1] += 6.0   # the end token comes early and often
    calls = []
    real_update = pol.update

    def spy(images, histories, actions, td_targets, apply_grads):
        calls.append(dict(n=len(actions), td=np.asarray(td_targets), apply=apply_grads, actions=np.asarray(actions),
                          frames=np.asarray(histories)))
        return real_update(images, histories, actions, td_targets, apply_grads)

    pol.update = spy
    np.random.seed(2)
    env = PolygonEnv(ds)
    rewards = reinforce(env, None, pol, total_episodes=11, max_timesteps_per_episode=T)
    assert len(rewards) == 11 and all(0.0 <= r <= 1.0 for r in rewards)
    earning = [k for k, r in enumerate(rewards) if r > 0]
    assert len(calls) == len(earning) and len(earning) >= 3
    for k, c in zip(earning, calls):
        assert c["apply"] == (k in (5, 10))
        assert c["actions"][-1].tolist() == [P, 0] and c["frames"].shape == (c["n"], P, P, 2)
        # the only reward is the terminal IoU: returns are its discounted tail
        assert np.allclose(c["td"], discount_rewards([0.0] * (c["n"] - 1) + [rewards[k]]))
    applied = sum(1 for k in earning if k in (5, 10))
    assert pol.global_step == applied


@pytest.mark.parametrize("B,T,P,end_token", [(3, 5, 28, False), (8, 10, 28, True), (1, 1, 8, False), (64, 5, 28, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sequence_loss_in_two_launches_against_the_oracle(B, T, P, end_token, dtype):
    """isg_sequence_xent_fwd (labels from (x, y) targets) + isg_sequence_summary (mean loss, the masked metrics and
    their quotients finished on the device) + isg_softmax_xent_mean_bwd (the mean's backward in one launch, upstream
    gradient read on the device): loss, gradient and metrics against the oracle's restatement of reference
    polyrnn/Model.py:39,44-54,98-100, and against the separate entry points they replace."""
    from imagesegmentation_amd.polyrnn import decode
    rs = np.random.RandomState(B * 100 + T)
    K = P * P + (1 if end_token else 0)
    z = rs.randn(B, T, K) * 2
    targets = rs.randint(0, P, [B, T, 2])
    if end_token:
        targets[0, T - 1] = (P, 0)  # the end token's class is P * P
    durations = rs.randint(0, T + 1, [B])
    durations[0] = T
    tz = torch.tensor(z, dtype=dtype, device=DEV).requires_grad_(True)
    zr = tz.detach().double().cpu().numpy()
    tt, td = torch.tensor(targets, dtype=torch.int32, device=DEV), torch.tensor(durations, dtype=torch.int32, device=DEV)
    loss, m = decode.sequence_loss(tz, tt, td, P)
    up = torch.tensor(0.5, device=DEV)  # an upstream gradient that is not one
    loss.backward(gradient=up)
    labels = targets[..., 0] * P + targets[..., 1]
    ref_rows = oc.softmax_xent(zr.reshape(B * T, K), labels.reshape(-1))
    assert abs(loss.item() - ref_rows.mean()) <= 2e-5 * max(1.0, abs(ref_rows.mean()))
    ref_grad = oc.softmax_xent_backward(zr.reshape(B * T, K), labels.reshape(-1), np.full([B * T], 0.5 / (B * T)))
    tol = 1e-6 if dtype == torch.float32 else 1e-2 * np.abs(ref_grad).max()
    assert np.abs(tz.grad.double().cpu().numpy().reshape(B * T, K) - ref_grad).max() <= max(tol, 1e-7)
    # metrics: the four sums equal isg_sequence_metrics on the kernel's own argmax, the quotients are theirs
    pred = decode.argmax_rows(tz.detach()).reshape(-1).to(torch.int32)
    m4 = decode.sequence_metrics(pred, tt, td, P).cpu().numpy()
    m6 = m.cpu().numpy()
    assert np.array_equal(m6[:4], m4)
    n = max(m4[3], 1e-300)
    if m4[3] > 0:
        assert m6[4] == m4[0] / n and m6[5] == m4[1] / n
    if m4[3] > 0:
        ref = oc.metrics(zr, targets, durations, P)
        assert abs(m6[4] - ref[0]) <= 1e-12 and abs(m6[5] - ref[1]) <= 1e-9 and abs(m6[2] - ref[2]) <= 1e-9
    # the separate entry points give the same mean and the same gradient
    tz2 = tz.detach().clone().requires_grad_(True)
    loss2 = decode.softmax_xent_mean(tz2, torch.tensor(labels, device=DEV))
    loss2.backward(gradient=up)
    assert abs(loss2.item() - loss.item()) <= 1e-6 * max(1.0, abs(loss.item()))
    assert torch.allclose(tz2.grad.float(), tz.grad.float(), rtol=0, atol=max(tol, 1e-7))


@pytest.mark.gpu
@pytest.mark.parametrize("hist", [0, 1, 2, 5])
def test_decode_begin_and_polygon_assembly_follow_the_host_loop(hist):
    """isg_decode_begin / isg_decode_polygons against the reference's own arithmetic (polyrnn/Model.py:218-230,
    252-260) written with numpy: the first history frames are the last `hist` true vertices (indices taken modulo the
    polygon's length, also where the polygon is shorter than the history), the closing vertex is frame 0's, and the
    rasterised polygon is those frames followed by the predicted vertices scaled to the image."""
    from imagesegmentation_amd.polyrnn import decode
    rs = np.random.RandomState(3 + hist)
    B, vmax, T = 9, 16, 6
    nvt = rs.randint(3, vmax + 1, [B]).astype(np.int32)
    nvt[0] = 3
    vt = rs.randint(0, 28, [B, vmax, 2]).astype(np.int32)
    d = "cuda:0"
    close, pts0, pts, required, durations, predicted = decode.decode_begin(torch.tensor(vt, device=d),
                                                                            torch.tensor(nvt, device=d), hist, T)
    want = np.zeros([B, max(hist, 1), 2], np.int32)
    for b in range(B):
        for i in range(max(hist, 1)):
            want[b, i] = vt[b, (int(nvt[b]) - hist + i) % int(nvt[b])]
    assert np.array_equal(close.cpu().numpy(), want[:, 0])
    assert np.array_equal(pts0.cpu().numpy(), want[:, :hist]) and np.array_equal(pts.cpu().numpy(), want[:, :hist])
    assert pts.data_ptr() != pts0.data_ptr() or hist == 0
    assert bool((required == 1).all()) and bool((durations == 0).all()) and bool((predicted == 0).all())
    assert tuple(predicted.shape) == (B, T, 2)

    pred = rs.randint(0, 28, [B, T, 2]).astype(np.int32)
    dur = rs.randint(0, T + 1, [B]).astype(np.int32)
    for prepend in (False, True):
        for by_duration in (False, True):
            vp, count = decode.decode_polygons(torch.tensor(pred, device=d), torch.tensor(dur, device=d), pts0, 8,
                                               prepend and hist > 0, by_duration, 24)
            lead = hist if prepend else 0
            ref = np.zeros([B, 24, 2], np.int32)
            ref[:, :lead] = want[:, :lead]
            ref[:, lead:lead + T] = pred * 8
            assert np.array_equal(vp.cpu().numpy(), ref)
            assert np.array_equal(count.cpu().numpy(), (dur if by_duration else np.full([B], T)) + lead)


@pytest.mark.gpu
def test_head_at_a_handful_of_rows_equals_the_library_gemm():
    """decode.dense_few_rows (reduction axis cut into batched GEMMs with float partial sums, one summing launch) against
    float64: no further from it than the library's one-GEMM result; shapes it does not take return None."""
    from imagesegmentation_amd.polyrnn import decode
    d = "cuda:0"
    g = torch.Generator(device=d).manual_seed(11)
    M, K, N = 16, 12544, 785
    flat = (torch.randn([M, K], device=d, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn([K, N], device=d, generator=g) * 0.01).to(torch.bfloat16)
    b = torch.randn([N], device=d, generator=g).to(torch.bfloat16)
    got = decode.dense_few_rows(flat, w, b)
    assert got is not None and got.dtype == torch.bfloat16 and tuple(got.shape) == (M, N)
    ref = flat.double() @ w.double() + b.double()
    one = torch.addmm(b, flat, w)
    err, err_one = float((got.double() - ref).abs().max()), float((one.double() - ref).abs().max())
    assert err <= max(err_one, 2.0 ** -8 * float(ref.abs().max()))
    assert decode.dense_few_rows(flat.float(), w.float(), b.float()) is None
    assert decode.dense_few_rows(flat[:, :1000].contiguous(), w[:1000].contiguous(), b) is None
    # a weight matrix padded with zero columns to a multiple of 8 (rows on 16-byte boundaries): the first N columns,
    # no further from float64 than the unpadded result
    wp = torch.zeros([K, 792], dtype=torch.bfloat16, device=d)
    wp[:, :N] = w
    padded = decode.dense_few_rows(flat, wp, b)
    assert padded is not None and tuple(padded.shape) == (M, N)
    assert float((padded.double() - ref).abs().max()) <= max(err_one, 2.0 ** -8 * float(ref.abs().max()))
    # rows that lie apart (one step of a sequence's kept outputs): the same bits as from a contiguous copy
    seq = torch.zeros([M, 3, K], dtype=torch.bfloat16, device=d)
    seq[:, 1] = flat
    apart = decode.dense_few_rows(seq[:, 1], w, b)
    assert apart is not None and torch.equal(apart, got)
    assert decode.dense_few_rows(seq[:, 1, 4:], w[4:], b) is None  # (rows that do not start on 16 bytes' multiples)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(20480, 785), (320, 785), (1, 5), (37, 1024), (5000, 33)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_column_sums_against_float64(rows, cols, dtype):
    """isg_column_sums (the dense head's bias gradient) against a float64 sum of the same values; two calls give the
    same bits (a fixed order of additions)."""
    from imagesegmentation_amd.polyrnn import decode
    g = torch.Generator(device="cuda:0").manual_seed(rows + cols)
    x = torch.randn([rows, cols], device="cuda:0", generator=g).to(dtype)
    got = decode.column_sums(x)
    ref = x.double().sum(0)
    assert got.dtype == torch.float32 and tuple(got.shape) == (cols,)
    assert float((got.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(x.double().abs().sum(0).max()))
    assert torch.equal(got, decode.column_sums(x))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,bias", [(1, 8, False), (5, 1000, True), (300, 32768, True)])
def test_one_column_dense_layer_against_float64(rows, cols, bias):
    """isg_rows_dot (a value head: one output column) against the float64 product of the same bf16 operands, and
    the float32 bias-less GEMM's answer for good measure; shapes it does not take answer None."""
    from imagesegmentation_amd.polyrnn import decode
    g = torch.Generator(device="cuda:0").manual_seed(rows * 7 + cols)
    x = torch.randn([rows, cols], device="cuda:0", generator=g).bfloat16()
    w = (torch.randn([cols, 1], device="cuda:0", generator=g) * 0.05).bfloat16()
    b = torch.randn([1], device="cuda:0", generator=g).bfloat16() if bias else None
    got = decode.rows_dot(x, w, b)
    assert got is not None and got.dtype == torch.float32 and tuple(got.shape) == (rows,)
    ref = (x.double() @ w.double())[:, 0] + (b.double() if bias else 0.0)
    scale = float((x.double().abs() @ w.double().abs()).max())
    assert float((got.double() - ref).abs().max()) <= 2e-6 * max(1.0, scale)
    assert torch.equal(got, decode.rows_dot(x, w, b))
    assert decode.rows_dot(x.float(), w.float(), None) is None
    assert decode.rows_dot(x[:, :cols - 4].contiguous(), w[:cols - 4].contiguous(), None) is None


@pytest.mark.gpu
@pytest.mark.parametrize("end_token", [False, True])
def test_update_that_adopts_the_rollouts_forward_pass_equals_the_update_that_runs_its_own(end_token):
    """BatchedReinforce.step with the recurrence of the rollout kept for the update (SequenceRecorder) against the
    update that runs the forward steps a second time (ISG_NO_FORWARD_REUSE): same episodes, and after two steps
    the same weights bit for bit - the adopted pass holds the numbers the second pass would compute, and what it
    holds beyond an episode's end never reaches a gradient."""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl.reinforce import BatchedReinforce, ExperimentPolicyEstimator
    P, T, N = 28, 10, 64
    ds = data.PolygonDataset.synthetic(32, P, P, T, 2, seed=5, shape_mask_fn=oracle_raster.create_shape_mask)
    runs = []
    for reuse in (True, False):
        pol = ExperimentPolicyEstimator(P, P, max_timesteps=T, end_token=end_token, compute_dtype=torch.bfloat16,
                                        device=DEV, sample_seed=11, seed=3)
        with torch.no_grad():
            pol.dense_bias[-1 if end_token else 0] += 4.0  # (episodes of different lengths, some ended early)
        br = BatchedReinforce(pol, ds, N, device_pool=True, pool_seed=8)
        assert br.reuse_forward
        br.reuse_forward = reuse
        eps = [br.step()[0] for _ in range(2)]
        runs.append((eps, [p.detach().clone() for p in pol.trainable_variables]))
    for a, b in zip(runs[0][0], runs[1][0]):
        for key in ("actions", "lengths", "reward", "points"):
            assert torch.equal(a[key], b[key]), key
    if end_token:  # (the biased end-token class ends episodes early: steps beyond an episode's end are in the pass)
        assert 0 < int(runs[0][0][0]["lengths"].min()) < T
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)
