"""CPU restatement of the reference's actor-critic specification (polyrl/a3c).
TEST INFRASTRUCTURE ONLY - the shipped package never imports it.

Parity status: the reference's a3c code never ran (see the package module's header) and holds no
tests or fixtures, so the network and loss below are "parity unpinned"; ``discount`` is pinned to
the library call the reference makes, scipy.signal.lfilter (polyrl/a3c/a3c.py:11-12).
"""
import numpy as np
import scipy.signal
import torch

from . import torch_port


def discount(x, gamma):
    """reference polyrl/a3c/a3c.py:11-12, the same call."""
    return scipy.signal.lfilter([1], [1, -gamma], np.asarray(x, np.float64)[::-1], axis=0)[::-1]


def process_rollout(rewards, values, bootstrap, gamma, lambda_=1.0):
    """reference polyrl/a3c/a3c.py:15-32 -> (advantages, returns)."""
    rewards = np.asarray(rewards, np.float64)
    vpred_t = np.asarray(list(values) + [bootstrap], np.float64)
    rewards_plus_v = np.asarray(list(rewards) + [bootstrap], np.float64)
    batch_r = discount(rewards_plus_v, gamma)[:-1]
    td_error = rewards + gamma * vpred_t[1:] - vpred_t[:-1]
    return discount(td_error, gamma * lambda_), batch_r


def policy_forward(tp, x, c, h):
    """reference polyrl/a3c/model.py:15-41 in float64 torch CPU ops.
    tp: dict(convs=[(w, b)] * 4, w_x, w_h, bias, action=(w, b), value=(w, b)); x [N,T,H,W,C].
    Returns logits [N,T,K], vf [N,T], (c, h)."""
    N, T = x.shape[:2]
    outs = []
    for t in range(T):
        y = x[:, t]
        for w, b in tp["convs"]:
            y = torch.relu(torch_port.conv_same(y, w, b))
        h, c = torch_port.cell(y, c, h, tp["w_x"], tp["w_h"], tp["bias"])
        outs.append(h)
    flat = torch.stack(outs, dim=1).reshape(N * T, -1)
    logits = flat @ tp["action"][0] + tp["action"][1]
    vf = (flat @ tp["value"][0] + tp["value"][1]).reshape(-1)
    return logits.reshape(N, T, -1), vf.reshape(N, T), (c, h)


def loss(logits, vf, actions, adv, r):
    """reference polyrl/a3c/a3c.py:201-216 for one rollout: logits [T,K], vf/adv/r [T], actions [T]."""
    log_prob = torch.log_softmax(logits, dim=1)
    prob = torch.softmax(logits, dim=1)
    ac = torch.nn.functional.one_hot(actions, logits.shape[1]).to(logits.dtype)
    pi_loss = -torch.sum(torch.sum(log_prob * ac, dim=1) * adv)
    vf_loss = 0.5 * torch.sum((vf - r) ** 2)
    entropy = -torch.sum(prob * log_prob)
    return pi_loss + 0.5 * vf_loss - entropy * 0.01
