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
    def reset(self, images=None, polygons=None):
        """Samples ``num_envs`` (image, polygon) pairs (``dataset.raw_sample``) unless given, and
        returns the state tensor [N, P, P, 3 + hist]."""
        if images is None:
            images, polygons = self.dataset.raw_sample(batch_size=self.num_envs)
        n, hist, size = self.num_envs, self.hist, self.size
        assert len(polygons) == n
        vmax_true = max(max(len(p) for p in polygons), util.VMAX_DEFAULT)
        self.vcap = max(vmax_true, hist + self.max_steps + 1)
        vt, nvt = util.pack_polygons(polygons, vmax=self.vcap)
        dev = self.device
        self.vertices = torch.from_numpy(vt).to(dev)
        self.num_vertices = torch.from_numpy(nvt).to(dev)
        self.predicted_vertices = torch.zeros([n, self.vcap, 2], dtype=torch.int32, device=dev)
        self.predicted_vertices[:, :hist] = self.vertices[:, :hist]
        self.num_predicted = torch.full([n], hist, dtype=torch.int32, device=dev)
        self.history_points = self.vertices[:, :hist].clone().contiguous()
        self.done = torch.zeros([n], dtype=torch.bool, device=dev)
        self.steps_taken = 0
        img = torch.as_tensor(np.asarray(images), device=dev).to(self.state_dtype)
        self._state = torch.zeros([n, size, size, 3 + hist], dtype=self.state_dtype, device=dev)
        self._state[..., :3] = img
        self._render_history()
        return self._state

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
    def step(self, actions, compute_reward=True):
        """actions: int tensor [N, 2] of (x, y).  Returns (state, reward float64 [N], done bool [N]).

        With ``compute_reward=False`` the terminal IoU is not evaluated in this call (reward is 0);
        use ``terminal_rewards()`` once all episodes have ended to get them in a single launch.
        """
        n = self.num_envs
        actions = torch.as_tensor(actions, device=self.device).to(torch.int32).reshape(n, 2)
        if self.steps_taken >= self.vcap - self.hist:
            raise RuntimeError("episode longer than the vertex buffer (max_steps=%d)" % self.max_steps)
        active = ~self.done
        first_true = self.vertices[:, 0]
        rows = torch.arange(n, device=self.device)
        slot = self.num_predicted.long()
        if self.end_token:
            is_end = active & (actions[:, 0] == self.size) & (actions[:, 1] == 0)
            appended = torch.where(is_end[:, None], first_true, actions)
            push = active & ~is_end
            terminal = is_end
        else:
            appended = actions
            push = active
            terminal = active & (actions == first_true).all(dim=1)
        current = self.predicted_vertices[rows, slot.clamp(max=self.vcap - 1)]
        self.predicted_vertices[rows, slot.clamp(max=self.vcap - 1)] = torch.where(active[:, None], appended, current)
        self.num_predicted = self.num_predicted + active.to(torch.int32)
        if self.hist > 0:
            rolled = torch.roll(self.history_points, shifts=1, dims=1)
            rolled[:, -1] = actions
            self.history_points = torch.where(push[:, None, None], rolled, self.history_points).contiguous()
            self._render_history()
        self.done = self.done | terminal
        self.steps_taken += 1
        reward = torch.zeros([n], dtype=torch.float64, device=self.device)
        if compute_reward and bool(terminal.any()):
            idx = terminal.nonzero(as_tuple=True)[0]
            inter, union = util.raster_iou(self.predicted_vertices[idx].contiguous(),
                                           self.num_predicted[idx].contiguous(), self.vertices[idx].contiguous(),
                                           self.num_vertices[idx].contiguous(), self.size)
            reward[idx] = inter.double() / union.double()
        return self._state, reward, self.done.clone()

    def terminal_rewards(self):
        """IoU of every environment's predicted polygon against its true one, in one launch:
        (reward float64 [N], inter int32 [N], union int32 [N]).  Only meaningful where done."""
        inter, union = util.raster_iou(self.predicted_vertices, self.num_predicted, self.vertices, self.num_vertices,
                                       self.size)
        return inter.double() / union.double(), inter, union
