"""REINFORCE on the polygon environments (reference polyrl/ReinforceOverfitExperiment.py and
polyrl/ReinforceOverfitEndTokenExperiment.py).

``ExperimentPolicyEstimator`` keeps the reference's surface - ``predict(image, history, rnn_state)``,
``update(images, histories, actions, td_targets, apply_grads)``, ``get_rnn_zero_state(batch_size)`` -
and its semantics: the History2Large network with P*P (+1) action logits, actions sampled from the
first time step's logits, loss = mean over [B, T] of cross-entropy(action) * td_target with padded
steps weighted 0, raw (unclipped) gradients summed into a buffer and applied by Adam when asked
(reference :111-134, :147-176).  ``reinforce`` is the reference's episode loop (:279-331).

``BatchedReinforce`` runs the same algorithm for N lock-stepped rollouts per GPU: one recurrent step
for all environments per launch, actions from isg_sample_actions, terminal IoU rewards of every rollout
from one isg_raster_iou launch, one teacher-forced update pass over the recorded episodes, gradients
all-reduced across ranks (RCCL) before Adam.  Episodes of one update are the batch: their
per-episode gradients are summed exactly as the reference sums them in its gradBuffer.
"""
import itertools
import math

import numpy as np
import torch

from .. import parallel
from ..convLSTM import LSTMStateTuple, RecurrentStepper, SequenceInput, SequenceRecorder, _run_sequence_forward
from ..polyrnn import decode
from ..polyrnn.Model import Model
from .batched_env import BatchedPolygonEnv, DevicePolygonPool


def discount_rewards(r, gamma=0.99):
    """Reverse cumulative discounted return (reference polyrl/MonteCarloPolicyGradient.py:13-22)."""
    r = np.asarray(r, dtype=np.float64)
    out = np.zeros_like(r)
    running = 0.0
    for t in reversed(range(r.size)):
        running = running * gamma + r[t]
        out[t] = running
    return out


class ExperimentPolicyEstimator(Model):
    encoder_spec = ((16, 5), (16, 5), (16, 3))
    rnn_layers = 2
    scope_name = 'policy_estimator'

    def __init__(self, image_size, action_size, tf_session=None, summary_writer=None, max_timesteps=10,
                 end_token=False, **kwargs):
        self.end_token = end_token
        self.action_size = action_size
        self.tf_session = tf_session
        self.summary_writer = summary_writer
        sample_seed = kwargs.pop("sample_seed", 0)
        self._sample_calls = 0
        super().__init__(tf_session, max_timesteps, image_size, action_size, 2, kwargs.pop("filepath", None), **kwargs)
        self._sample_base, self._sample_rank_seed = sample_seed, None
        self.gradBuffer = torch.zeros_like(self._flat_grad)

    @property
    def _sample_seed(self):
        """Per-rank action noise, folded at the first sampled action (a process group initialised after the
        estimator was built is then already there)."""
        if self._sample_rank_seed is None:
            self._sample_rank_seed = parallel.sampling_seed(self._sample_base, self.process_group)
        return self._sample_rank_seed

    # -- acting ----------------------------------------------------------------------------------
    def get_rnn_zero_state(self, batch_size):
        return self.rnn.zero_state(batch_size, self.device)

    def step_logits(self, features, history_points, rnn_state):
        """One recurrent step for a batch: features [B,P,P,Cf], history_points int [B,hist,2].
        Returns (logits [B,K], new state)."""
        seq = SequenceInput(static=features, hist_points=history_points[:, None].contiguous())
        out, final, _ = _run_sequence_forward(list(self.rnn.cells), seq, None, 1, [(s.c, s.h) for s in rnn_state], False)
        state = tuple(LSTMStateTuple(c.clone(), h.clone()) for c, h in final)
        return self.head(out)[:, 0], state

    def sample_actions(self, logits):
        """Categorical sample per row -> (x, y) = (a // P, a % P); the end token is class P*P = (P, 0)."""
        a = decode.sample_actions(logits, self._sample_seed, self.action_size,
                                  row_offset=self._sample_calls * (1 << 20))
        self._sample_calls += 1
        return a

    def predict(self, image, history, rnn_state):
        """Picks a single action given image [P,P,3] and history frames [P,P,hist] (reference :136-145).
        Returns (action [2], new rnn state)."""
        with torch.no_grad():
            features = self.encode(np.asarray(image)[None])
            pts = np.full([1, self.history_length, 2], -1, np.int32)
            h = np.asarray(history)
            for i in range(self.history_length):
                ys, xs = np.nonzero(h[:, :, i])
                if len(xs):
                    pts[0, i] = (xs[0], ys[0])
            logits, state = self.step_logits(features, torch.as_tensor(pts, device=self.device), rnn_state)
            action = self.sample_actions(logits)[0].cpu().numpy()
        return action, state

    # -- learning ----------------------------------------------------------------------------------
    def policy_loss(self, images, durations, histories, actions, td_targets, recorded=None):
        """mean over [B, T] of xent(action) * td (reference :45-48).  ``recorded``: see ``Model.forward``."""
        logits, _ = self.forward(images, durations, histories, recorded=recorded)
        actions = self._to_dev(actions, torch.int64)
        dense = actions[..., 0] * self.action_size + actions[..., 1]
        td = self._to_dev(td_targets, torch.float32)
        return decode.softmax_xent_rows(logits, dense, td).mean()

    def apply_accumulated(self, all_reduce=True):
        """Adam on the summed raw gradients (reference ``update_batch``, :130-131, :171-177)."""
        self._flat_grad.copy_(self.gradBuffer)
        self._apply_gradients(clip=False, all_reduce=all_reduce, average=False)
        self.gradBuffer.zero_()

    def update(self, images, histories, actions, td_targets, apply_grads):
        """One episode (reference :147-177): images [P,P,3], histories [n,P,P,hist], actions [n,2],
        td_targets [n]; pads to max_timesteps, accumulates the gradient, applies it when asked."""
        T, P, hist = self.max_timesteps, self.action_size, self.history_length
        n = np.asarray(histories).shape[0]
        td = np.concatenate([np.asarray(td_targets, np.float64), np.zeros([T - n])])[None]
        at = np.concatenate([np.asarray(actions), np.zeros([T - n, 2])], axis=0)[None]
        h = np.concatenate([np.asarray(histories), np.zeros([T - n, P, P, hist])], axis=0)[None]
        self._flat_grad.zero_()
        loss = self.policy_loss(np.asarray(images)[None], np.array([n]), h, at, td)
        loss.backward()
        self.gradBuffer += self._flat_grad
        if apply_grads:
            self.apply_accumulated()
        return float(loss.detach())


class Trajectory:
    """What one scalar episode leaves behind for the update: the image, and per step the history frames the
    policy saw, the action it took and the reward it got."""

    def __init__(self, image):
        self.image = image
        self.histories, self.actions, self.rewards = [], [], []

    def add(self, history, action, reward):
        self.histories.append(history)
        self.actions.append(action)
        self.rewards.append(reward)

    @property
    def total_reward(self):
        return sum(self.rewards)

    @property
    def final_reward(self):
        return self.rewards[-1] if self.rewards else 0


def play_episode(env, policy_estimator, max_timesteps=None):
    """Resets ``env`` and follows the policy until the episode ends or ``max_timesteps`` actions were taken."""
    observation = env.reset()
    trajectory = Trajectory(observation[:, :, :3])
    rnn_state = policy_estimator.get_rnn_zero_state(batch_size=1)
    budget = itertools.count() if max_timesteps is None else range(max_timesteps)
    for _ in budget:
        frames = observation[:, :, 3:]
        action, rnn_state = policy_estimator.predict(image=trajectory.image, history=frames, rnn_state=rnn_state)
        observation, reward, done = env.step(action)
        trajectory.add(frames, action, reward)
        if done:
            break
    return trajectory


def validate_episode_iou(policy_estimator, env, max_timesteps_per_episode=None, summary_prefix='valid'):
    """The estimator's environment-driven validation (reference polyrl/ReinforceOverfitExperiment.py:224-261): one
    episode under the sampling policy, then the IoU of the polygon [last ``hist`` true vertices ++ the actions
    taken] against the true polygon.  Returns the IoU (also kept as ``policy_estimator.last_ious`` and written to
    the estimator's scalar log under ``<prefix>/iou``)."""
    from .. import util
    trajectory = play_episode(env, policy_estimator, max_timesteps_per_episode)
    true_vertices = np.asarray(env.vertices)
    hist = policy_estimator.history_length
    predicted = np.concatenate([true_vertices[len(true_vertices) - hist:], np.vstack(trajectory.actions)], axis=0)
    P = policy_estimator.action_size
    # (an end-token action (P, 0) is a vertex like any other here, as in the reference, whose mask[rr, cc] = 1 then
    # raises IndexError for column P - util.calculate_iou raises the same)
    iou = util.calculate_iou(predicted.astype(np.int64), true_vertices.astype(np.int64), P)
    policy_estimator.last_ious = np.array([iou])
    tag = summary_prefix + '/iou' if summary_prefix else 'iou'
    policy_estimator._add_summaries(summary_prefix or 'valid', extra={tag: float(iou)})
    return float(iou)


def reinforce(*args, **kwargs):
    """The reference's scalar training loop in both of its call shapes:

    * ``reinforce(env, sess, policy_estimator, total_episodes=None, max_timesteps_per_episode=None,
      summary_writer=None)`` - polyrl/ReinforceOverfitEndTokenExperiment.py:277 (called by keyword at its :352);
    * ``reinforce(train_env, valid_env, sess, policy_estimator, total_episodes=None,
      max_timesteps_per_episode=None, summary_writer=None)`` - polyrl/ReinforceOverfitExperiment.py:279 (called by
      keyword at its :358).

    Episode after episode; an episode whose last step earned a reward (:306) adds its policy gradient, with
    discounted returns as weights, to the estimator's buffer; the buffer is applied on every fifth episode but
    the first (:312).  ``validate_every=n`` (keyword, not in the reference, default off) runs the reference's
    validation episodes - on the training environment, and on ``valid_env`` when there is one - on every n-th
    episode from episode 0 on (the reference's n is 100, its :326-331); they draw from ``np.random`` through
    ``env.reset`` like the reference's.  ``sess`` and ``summary_writer`` are accepted for the call shape only.
    Returns the list of episode rewards."""
    names = ("env", "sess", "policy_estimator", "total_episodes", "max_timesteps_per_episode", "summary_writer")
    two_envs = "train_env" in kwargs or "valid_env" in kwargs or \
        (len(args) >= 2 and hasattr(args[1], "reset") and hasattr(args[1], "step"))
    if two_envs:
        names = ("train_env", "valid_env") + names[1:]
    names += ("validate_every",)
    if len(args) > len(names):
        raise TypeError("reinforce() takes at most %d positional arguments (%d given)" % (len(names), len(args)))
    bound = dict(zip(names, args))
    for k, v in kwargs.items():
        if k not in names:
            raise TypeError("reinforce() got an unexpected keyword argument %r" % k)
        if k in bound:
            raise TypeError("reinforce() got multiple values for argument %r" % k)
        bound[k] = v
    for k in names[:4 if two_envs else 3]:
        if k not in bound:
            raise TypeError("reinforce() missing required argument %r" % k)
    env = bound["train_env"] if two_envs else bound["env"]
    valid_env = bound.get("valid_env")
    policy_estimator = bound["policy_estimator"]
    total_episodes = bound.get("total_episodes")
    max_timesteps_per_episode = bound.get("max_timesteps_per_episode")
    validate_every = bound.get("validate_every")
    episode_rewards = []
    numbering = itertools.count() if total_episodes is None else range(total_episodes)
    for episode in numbering:
        trajectory = play_episode(env, policy_estimator, max_timesteps_per_episode)
        if trajectory.final_reward > 0:
            policy_estimator.update(images=trajectory.image, histories=np.stack(trajectory.histories, axis=0),
                                    actions=np.vstack(trajectory.actions),
                                    td_targets=discount_rewards(trajectory.rewards),
                                    apply_grads=episode > 0 and episode % 5 == 0)
        episode_rewards.append(trajectory.total_reward)
        if validate_every and episode % validate_every == 0:
            validate_episode_iou(policy_estimator, env, max_timesteps_per_episode,
                                 summary_prefix='train' if two_envs else 'valid')
            if valid_env is not None:
                validate_episode_iou(policy_estimator, valid_env, max_timesteps_per_episode, summary_prefix='valid')
    return episode_rewards


class BatchedReinforce:
    """N lock-stepped REINFORCE rollouts per GPU against a BatchedPolygonEnv."""

    def __init__(self, policy, dataset, num_envs, gamma=0.99, device_pool=False, pool_seed=0):
        """``device_pool=True`` keeps the dataset on the device (``DevicePolygonPool``): rollouts that are
        not handed explicit images / polygons then draw their episodes there instead of through
        ``dataset.raw_sample`` on the host."""
        self.policy = policy
        self.env = BatchedPolygonEnv(dataset, num_envs, end_token=policy.end_token, max_steps=policy.max_timesteps,
                                     device=policy.device, state_dtype=torch.float32)
        self.num_envs = num_envs
        self.gamma = gamma
        import os
        dt_ok = policy.compute_dtype == torch.bfloat16 and policy.device.type == "cuda"
        #: the update adopts the rollout's forward pass of the recurrence (bf16 on the GPU: both passes then run the
        #: same kernels on the same inputs; elsewhere the update runs its own)
        self.reuse_forward = dt_ok and not os.environ.get("ISG_NO_FORWARD_REUSE")
        self.pool = None
        if device_pool:
            self.pool = DevicePolygonPool(dataset, policy.device, dtype=torch.float32,
                                          vcap=policy.history_length + policy.max_timesteps + 1, seed=pool_seed)

    def rollout(self, images=None, polygons=None, keep_forward=False):
        """Runs every environment for up to max_timesteps sampled actions.  Returns a dict with the
        recorded episode (history points, actions, lengths), rewards (float64 [N]) and counts.
        ``keep_forward``: the recurrence keeps what its backward sweep reads (``SequenceRecorder``), so that the
        update of these episodes does not run the same forward steps a second time (entry "forward")."""
        pol, env = self.policy, self.env
        N, T, hist = self.num_envs, pol.max_timesteps, pol.history_length
        if images is None and self.pool is not None:
            env.reset(device_batch=self.pool.sample(N))
        else:
            env.reset(images=images, polygons=polygons)
        dev = pol.device
        points = torch.full([N, T, hist, 2], -1, dtype=torch.int32, device=dev)
        actions = torch.zeros([N, T, 2], dtype=torch.int32, device=dev)
        lengths = torch.zeros([N], dtype=torch.int32, device=dev)
        with torch.no_grad():
            features = pol.encode(env.image)
            stepper = SequenceRecorder(pol.rnn, features, hist, T) if keep_forward else \
                RecurrentStepper(pol.rnn, features, hist)
            for t in range(T):
                logits = pol.head(stepper.step(env.history_points))[:, 0]
                a = pol.sample_actions(logits)
                # the env kernel also files the step's history points, action and length
                env.advance(a, record=(points, actions, lengths), t=t)
            reward, inter, union = env.terminal_rewards()
            # an episode that never terminated earns nothing (the scalar loop ends it with reward 0)
            reward = torch.where(env.done, reward, torch.zeros_like(reward))
        return dict(images=env.image, points=points, actions=actions, lengths=lengths, reward=reward, inter=inter,
                    union=union, done=env.done.clone(), forward=stepper.record() if keep_forward else None)

    def returns(self, ep):
        """Discounted return per step: the only non-zero reward is the terminal one, so
        G_t = gamma^(len-1-t) * reward for t < len, 0 after."""
        T = self.policy.max_timesteps
        t = torch.arange(T, device=self.policy.device)[None, :]
        n = ep["lengths"][:, None]
        g = torch.pow(torch.tensor(self.gamma, dtype=torch.float64, device=self.policy.device),
                      (n - 1 - t).clamp(min=0).double()) * ep["reward"][:, None]
        return torch.where(t < n, g, torch.zeros_like(g)).float()

    def update(self, ep):
        """One accumulate-and-apply over all episodes of the batch (sum of per-episode gradients)."""
        pol = self.policy
        pol._flat_grad.zero_()
        td = self.returns(ep)
        loss = pol.policy_loss(ep["images"], ep["lengths"], ep["points"], ep["actions"], td,
                               recorded=ep.get("forward")) * self.num_envs
        ep["forward"] = None  # (its buffers belong to the update's backward sweep, which releases them)
        loss.backward()
        pol.gradBuffer += pol._flat_grad
        pol.apply_accumulated()
        return loss.detach() / self.num_envs

    def step(self, images=None, polygons=None):
        # (the weights do not change between a step's rollout and its update: the update adopts the rollout's
        # forward pass; ISG_NO_FORWARD_REUSE=1 runs it again, for A/B runs)
        ep = self.rollout(images, polygons, keep_forward=self.reuse_forward)
        loss = self.update(ep)
        return ep, loss
