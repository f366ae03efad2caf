"""N polygon environments stepped in lock-step on one GPU.

Same episode semantics as the scalar environments (reference polyrl/polygon_env.py:12-53 and
polyrl/polygon_env_end_token.py:12-54), but every environment's vertices, history points and flags
live in device tensors, history frames are rendered by ``isg_history_frames`` and the terminal
rewards of all environments that finish in a step come from ONE ``isg_raster_iou`` launch.

Differences from N scalar environments, by construction of a batch:
  * an environment that is already done ignores further actions (reward 0, done stays True)
    instead of raising;
  * rewards are float64 tensors, ``intersection / union`` evaluated in IEEE double on the device
    (bit-identical to Python's int / int).
"""
import numpy as np
import torch

from .. import _lib, util


class DevicePolygonPool:
    """A dataset's images and polygons resident on the device, and ``raw_sample``'s draw (reference
    Dataset.py:152-162: a batch of distinct indices, each polygon rolled by a random start vertex) as
    tensor operations: resetting N environments moves nothing over PCIe and runs no Python loop.

    ``sample(n)`` -> (images [n,S,S,3] in ``dtype``, vertices int32 [n,vcap,2] (rolled, zero past the
    count), counts int32 [n], indices int64 [n]).  Indices are distinct while n <= len(dataset) (as
    ``np.random.choice(..., replace=False)``) and drawn with replacement beyond that.  The stream is
    torch's device generator, not numpy's: same distribution, different numbers."""

    def __init__(self, dataset, device, dtype=torch.float32, vcap=util.VMAX_DEFAULT, seed=0):
        polys = [np.asarray(v) for v in dataset._vertices]
        self.vcap = max(int(vcap), max(len(v) for v in polys))
        vt, nvt = util.pack_polygons(polys, vmax=self.vcap)
        self.device = torch.device(device)
        self.images = torch.as_tensor(np.asarray(dataset._images), device=self.device).to(dtype)
        self.vertices = torch.from_numpy(vt).to(self.device)
        self.counts = torch.from_numpy(nvt).to(self.device)
        self.generator = torch.Generator(device=self.device)
        self.generator.manual_seed(int(seed))
        self._slot = torch.arange(self.vcap, device=self.device, dtype=torch.int64)[None, :]

    def __len__(self):
        return self.images.shape[0]

    def sample(self, n):
        m = len(self)
        if n <= m:
            idx = torch.randperm(m, generator=self.generator, device=self.device)[:n]
        else:
            idx = torch.randint(0, m, [n], generator=self.generator, device=self.device)
        cnt = self.counts[idx].to(torch.int64)
        start = (torch.rand([n], generator=self.generator, device=self.device, dtype=torch.float64) * cnt).long()
        start = torch.minimum(start, cnt - 1)
        # np.roll(p, start): rolled[j] = p[(j - start) mod count]
        src = torch.remainder(self._slot - start[:, None], cnt[:, None])
        rolled = torch.gather(self.vertices[idx], 1, src[:, :, None].expand(-1, -1, 2))
        rolled = torch.where((self._slot < cnt[:, None])[:, :, None], rolled, torch.zeros_like(rolled))
        return self.images[idx], rolled.contiguous(), cnt.to(torch.int32), idx


class BatchedPolygonEnv:
    def __init__(self, dataset, num_envs, end_token=False, max_steps=10, device=None, state_dtype=torch.float32):
        self.dataset = dataset
        self.num_envs = int(num_envs)
        self.end_token = bool(end_token)
        self.max_steps = int(max_steps)
        self.device = torch.device(device) if device is not None else util._device()
        self.state_dtype = state_dtype
        self.size = int(dataset._prediction_size)
        self.hist = int(dataset._history_length)

    # -- set-up ----------------------------------------------------------------------------------
    def reset(self, images=None, polygons=None, device_batch=None):
        """Samples ``num_envs`` (image, polygon) pairs (``dataset.raw_sample``) unless given, and
        returns the state tensor [N, P, P, 3 + hist].  ``device_batch`` = (images, vertices int32
        [N,V,2], counts int32 [N]) device tensors, e.g. from ``DevicePolygonPool.sample``, starts the
        episodes without touching the host."""
        n, hist = self.num_envs, self.hist
        dev = self.device
        if device_batch is not None:
            img, vt, nvt = device_batch[:3]
            assert vt.shape[0] == n and nvt.shape[0] == n and vt.dtype == torch.int32 and nvt.dtype == torch.int32
            need = hist + self.max_steps + 1
            if vt.shape[1] < need:
                vt = torch.nn.functional.pad(vt, (0, 0, 0, need - vt.shape[1]))
            self.vcap = int(vt.shape[1])
            self.vertices, self.num_vertices = self._keep("vertices", vt.contiguous()), self._keep("num_vertices", nvt.contiguous())
            img = img.to(self.state_dtype)
        else:
            if images is None:
                images, polygons = self.dataset.raw_sample(batch_size=n)
            assert len(polygons) == n
            vmax_true = max(max(len(p) for p in polygons), util.VMAX_DEFAULT)
            self.vcap = max(vmax_true, hist + self.max_steps + 1)
            vt, nvt = util.pack_polygons(polygons, vmax=self.vcap)
            self.vertices = self._keep("vertices", torch.from_numpy(vt).to(dev))
            self.num_vertices = self._keep("num_vertices", torch.from_numpy(nvt).to(dev))
            img = torch.as_tensor(np.asarray(images), device=dev).to(self.state_dtype)
        size = self.size
        # (every tensor of the episode keeps its address from reset to reset while the shapes stay the same: a
        # rollout captured as a graph reads and writes them in place)
        self.predicted_vertices = self._keep("predicted_vertices", None, [n, self.vcap, 2], torch.int32).zero_()
        self.predicted_vertices[:, :hist] = self.vertices[:, :hist]
        self.num_predicted = self._keep("num_predicted", None, [n], torch.int32).fill_(hist)
        self.history_points = self._keep("history_points", self.vertices[:, :hist].contiguous())
        self.done = self._keep("done", None, [n], torch.bool).zero_()
        self.steps_taken = 0
        self._state = self._keep("_state", None, [n, size, size, 3 + hist], self.state_dtype)
        self._state[..., :3] = img
        if hist:
            self._state[..., 3:] = 0
        self._render_history()
        return self._state

    def _keep(self, name, value, shape=None, dtype=None):
        """The tensor this environment already holds under ``name`` when it has the wanted shape and type (filled
        with ``value`` when one is given), else a new one."""
        old = getattr(self, name, None)
        if value is not None:
            shape, dtype = value.shape, value.dtype
        if old is not None and tuple(old.shape) == tuple(shape) and old.dtype == dtype and old.device == self.device:
            return old.copy_(value) if value is not None else old
        return value.clone() if value is not None else torch.empty(list(shape), dtype=dtype, device=self.device)

    @property
    def image(self):
        return self._state[..., :3]

    @property
    def histories(self):
        return self._state[..., 3:]

    def _render_history(self):
        if self.hist == 0:
            return
        code = {torch.float32: 0, torch.bfloat16: 1}[self.state_dtype]
        rc = _lib.lib().isg_history_frames(_lib.ptr(self.history_points), self.num_envs, self.size, self.hist, code,
                                           _lib.ptr(self._state), 3 + self.hist, 3, _lib.current_stream())
        _lib.check(rc, "isg_history_frames")

    # -- stepping --------------------------------------------------------------------------------
    def advance(self, actions, record=None, t=0):
        """One ``isg_env_step`` launch for every environment and nothing else: no reward, no rendered
        state, no allocation.  actions: int32 device tensor [N, 2] of (x, y).  ``record`` as in ``step``."""
        n = self.num_envs
        if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
            actions = torch.as_tensor(actions, device=self.device).to(torch.int32).contiguous()
        assert actions.numel() == 2 * n
        if self.steps_taken >= self.vcap - self.hist:
            raise RuntimeError("episode longer than the vertex buffer (max_steps=%d)" % self.max_steps)
        rec_p, rec_a, rec_l, T = None, None, None, 1
        if record is not None:
            rec_p, rec_a, rec_l = record
            T = rec_a.shape[1]
            assert (rec_p is None or rec_p.is_contiguous()) and rec_a.is_contiguous() and rec_l.is_contiguous()
        rc = _lib.lib().isg_env_step(_lib.ptr(actions), _lib.ptr(self.vertices), self.vcap * 2,
                                     _lib.ptr(self.predicted_vertices), self.vcap, _lib.ptr(self.num_predicted),
                                     _lib.ptr(self.history_points), self.hist, _lib.ptr(self.done), n, self.size,
                                     int(self.end_token), _lib.ptr(rec_p), _lib.ptr(rec_a), _lib.ptr(rec_l), int(T),
                                     int(t), _lib.current_stream())
        _lib.check(rc, "isg_env_step")
        self.steps_taken += 1

    def step(self, actions, compute_reward=True, render=True, record=None, t=0, host_sync=True):
        """actions: int tensor [N, 2] of (x, y).  Returns (state, reward float64 [N], done bool [N]).
        ``host_sync=False``: the terminal IoU is evaluated for every environment in one launch and masked on the
        device, instead of asking the host which environments finished - same rewards, no round trip.

        One ``isg_env_step`` launch updates every environment.  With ``compute_reward=False`` the
        terminal IoU is not evaluated in this call (reward is 0); use ``terminal_rewards()`` once all
        episodes have ended to get them in a single launch.  ``render=False`` skips the history frames
        of the returned state tensor (a policy that takes the history as points does not need them).
        ``record`` = (points int32 [N,T,hist,2], actions int32 [N,T,2], lengths int32 [N]) receives the
        step's episode record at slot ``t``.
        """
        n = self.num_envs
        actions = torch.as_tensor(actions, device=self.device).to(torch.int32).reshape(n, 2).contiguous()
        before = self.done.clone() if compute_reward else None
        self.advance(actions, record, t)
        if render and self.hist > 0:
            self._render_history()
        reward = torch.zeros([n], dtype=torch.float64, device=self.device)
        if compute_reward and not host_sync:
            terminal = self.done & ~before
            inter, union = util.raster_iou(self.predicted_vertices, self.num_predicted, self.vertices,
                                           self.num_vertices, self.size)
            reward = torch.where(terminal, inter.double() / union.double(), reward)
        elif compute_reward:
            terminal = self.done & ~before
            if bool(terminal.any()):
                idx = terminal.nonzero(as_tuple=True)[0]
                inter, union = util.raster_iou(self.predicted_vertices[idx].contiguous(),
                                               self.num_predicted[idx].contiguous(), self.vertices[idx].contiguous(),
                                               self.num_vertices[idx].contiguous(), self.size)
                reward[idx] = inter.double() / union.double()
        return self._state, reward, self.done.clone()

    def step_recorded(self, actions, t, rec_actions, rec_lengths, rec_rewards, step_values=None, rec_values=None,
                      render=True):
        """``step(host_sync=False)`` for a rollout that files its record on the device, in four launches (environment
        step, history frames, IoU, record) and without the tensor operations between them: slot ``t`` of
        ``rec_actions`` int32 [N, T, 2] gets the action, ``rec_lengths`` int32 [N] - zero at the rollout's first
        step - counts the steps an environment took while running, ``rec_rewards`` float64 [N, T] gets the IoU where
        the environment ended at this step and 0 elsewhere, ``rec_values`` float32 [N, T] the ``step_values`` [N].
        Returns the next state."""
        n = self.num_envs
        self.advance(actions, (None, rec_actions, rec_lengths), t)
        if render and self.hist > 0:
            self._render_history()
        inter, union = util.raster_iou(self.predicted_vertices, self.num_predicted, self.vertices, self.num_vertices,
                                       self.size)
        T = rec_actions.shape[1]
        assert rec_rewards.dtype == torch.float64 and rec_rewards.is_contiguous() and tuple(rec_rewards.shape) == (n, T)
        assert rec_lengths.dtype == torch.int32
        if rec_values is not None:
            assert rec_values.dtype == torch.float32 and rec_values.is_contiguous() and tuple(rec_values.shape) == (n, T)
            assert step_values.dtype == torch.float32 and step_values.is_contiguous() and step_values.numel() == n
        rc = _lib.lib().isg_rollout_record(_lib.ptr(self.done), _lib.ptr(rec_lengths), _lib.ptr(inter), _lib.ptr(union),
                                           _lib.ptr(step_values) if rec_values is not None else None, n, T, int(t),
                                           _lib.ptr(rec_rewards), _lib.ptr(rec_values), _lib.current_stream())
        _lib.check(rc, "isg_rollout_record")
        return self._state

    def terminal_rewards(self):
        """IoU of every environment's predicted polygon against its true one, in one launch:
        (reward float64 [N], inter int32 [N], union int32 [N]).  Only meaningful where done."""
        inter, union = util.raster_iou(self.predicted_vertices, self.num_predicted, self.vertices, self.num_vertices,
                                       self.size)
        return inter.double() / union.double(), inter, union
