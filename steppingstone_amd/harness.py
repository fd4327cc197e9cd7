"""Device-resident counterparts of the small harness pieces that sit on either side of env.step() in the
reference's PPO loop: GAE / returns, the left-right mirror augmentation, LR schedules.  Each is
pinned by golden vectors produced by the reference's own code (tests/golden/harness_golden.npz,
tools/make_golden.py).  All functions take and return torch tensors on whatever device the inputs live on.
"""
import torch


def compute_returns(rewards, value_preds, masks, bad_masks, next_value, use_gae=True, gamma=0.99, gae_lambda=0.95):
    """algorithms/storage.py:59-82.  rewards [T,N,1]; value_preds/masks/bad_masks [T+1,N,1]; returns [T+1,N,1]."""
    T = rewards.shape[0]
    returns = torch.zeros_like(value_preds)
    if use_gae:
        value_preds = value_preds.clone()
        value_preds[-1] = next_value
        gae = torch.zeros_like(next_value)
        for step in reversed(range(T)):
            delta = rewards[step] + gamma * value_preds[step + 1] * masks[step + 1] - value_preds[step]
            gae = delta + gamma * gae_lambda * masks[step + 1] * gae
            gae = gae * bad_masks[step + 1]
            returns[step] = gae + value_preds[step]
    else:
        returns[-1] = next_value
        for step in reversed(range(T)):
            returns[step] = ((returns[step + 1] * gamma * masks[step + 1] + rewards[step]) * bad_masks[step + 1]
                             + (1 - bad_masks[step + 1]) * value_preds[step])
    return returns


def mirror_batch(obs, act, indices):
    """common/envs_utils.py:687-740 restricted to what changes: returns (cat[obs, mirrored obs], cat[act, mirrored])."""
    neg_o, right_o, left_o, neg_a, right_a, left_a = [torch.as_tensor(i, dtype=torch.long, device=obs.device) for i in indices]

    def mirrored(t, neg, r, l):
        m = t.clone()
        m[:, neg] = -m[:, neg]
        rl, lr = torch.cat([r, l]), torch.cat([l, r])
        m[:, rl] = m[:, lr]
        return m

    return torch.cat([obs, mirrored(obs, neg_o, right_o, left_o)]), torch.cat([act, mirrored(act, neg_a, right_a, left_a)])


def get_mirror_function(indices):
    """Drop-in for common.envs_utils.get_mirror_function: same 8-tuple in, same 8-tuple out."""

    def mirror_function(sample):
        obs, states, act, value_preds, returns, masks, old_logp, adv = sample
        obs2, act2 = mirror_batch(obs, act, indices)
        rep = lambda t: t.repeat((2, 1))  # noqa: E731
        return obs2, rep(states), act2, rep(value_preds), rep(returns), rep(masks), rep(old_logp), rep(adv)

    return mirror_function


def linear_decay(epoch, total_num_epochs, initial_value, final_value):
    """common/misc_utils.py:20-23"""
    return initial_value - (initial_value - final_value) * epoch / float(total_num_epochs)


def exponential_decay(epoch, rate, initial_value, final_value):
    """common/misc_utils.py:26-27"""
    return max(initial_value * (rate ** epoch), final_value)
