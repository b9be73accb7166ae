"""Shared bodies (tests only) for the device-tier camera / frame-stacking path: ``SyntheticVecEnv(pixel=...,
frame_stacks=n)`` -- raw uint8 frames rendered and stored once per step on the device, the stacked observation and the
stacked sub-trajectory windows produced by one gather (smx_frame_stack_u8) -- against the HOST statement of the same
thing: ``SyntheticEnv`` (one actor, numpy) under ``FrameStackWrapper`` (surreal/env/wrapper.py:407-472; pinned to the
reference's wrapper by tests/golden/envwrap.json in tests/test_env_adapters.py).  Bit-exact: bytes and fp32 state.
CPU tier: torch-CPU kernel double; GPU tier: the HIP kernels."""
import numpy as np
import torch

from surreal_amd.env import FrameStackWrapper, SyntheticEnv, SyntheticVecEnv, stack_sources
from surreal_amd.env.exp_sender_wrapper import window_advance, windows_per_episode
from surreal_amd.session import Config


def check_device_camera_matches_host_framestack(n=5, D=7, A=3, pixel=(2, 20, 24), stacks=3, T=7, n_step=3, stride=2):
    rs = np.random.RandomState(3)
    actions = rs.uniform(-1.2, 1.2, size=(T, n, A)).astype(np.float32)
    # ---- host: one wrapped env per actor ----------------------------------------------------------------------
    cfg = Config(frame_stacks=stacks, frame_stack_concatenate_on_env=True)
    host_pix = np.zeros((n, T + 1, stacks * pixel[0]) + pixel[1:], np.uint8)
    host_low = np.zeros((n, T + 1, D), np.float32)
    host_rew, host_done = np.zeros((n, T), np.float32), np.zeros((n, T), np.float32)
    for i in range(n):
        env = FrameStackWrapper(SyntheticEnv(D, A, episode_len=T, seed=i, pixel=pixel), cfg)
        assert tuple(env.observation_spec()['pixel']['camera0']) == (stacks * pixel[0],) + pixel[1:]
        obs, _ = env.reset()
        for t in range(T + 1):
            host_pix[i, t], host_low[i, t] = obs['pixel']['camera0'], obs['low_dim']['flat_inputs']
            if t == T:
                break
            obs, r, d, _ = env.step(actions[t, i])
            host_rew[i, t], host_done[i, t] = r, float(d)
    # ---- device: all actors at once ---------------------------------------------------------------------------
    venv = SyntheticVecEnv(n, D, A, episode_len=T, seeds=list(range(n)), pixel=pixel, frame_stacks=stacks)
    dev = venv.device
    venv.start_rollout(T, info_width=2 * A)
    for t in range(T):
        obs = venv.observation()
        got = obs['pixel']['camera0']
        assert got.dtype == torch.uint8 and tuple(got.shape) == (n, stacks * pixel[0]) + pixel[1:]
        np.testing.assert_array_equal(got.cpu().numpy(), host_pix[:, t], err_msg='stacked camera observation, step %d' % t)
        np.testing.assert_array_equal(obs['low_dim']['flat_inputs'].cpu().numpy(), host_low[:, t])
        venv.step(torch.as_tensor(actions[t]).to(dev), pds=torch.zeros(n, 2 * A, device=dev))
    assert venv.slot == T
    np.testing.assert_array_equal(venv.rolls['rewards'][:, :T].cpu().numpy(), host_rew)
    np.testing.assert_array_equal(venv.rolls['dones'][:, :T].cpu().numpy(), host_done)
    # one RAW frame per step is what the device keeps (not `stacks` copies of it)
    assert tuple(venv.frames.shape) == (n, T + 1) + pixel
    np.testing.assert_array_equal(venv.frames[:, :, :].cpu().numpy(), host_pix[:, :, -pixel[0]:])
    # ---- the stacked windows (moving-window rule of exp_sender_wrapper.py:209-228 + stacking, one gather) --------
    W, adv = windows_per_episode(T, n_step, stride), window_advance(n_step, stride)
    f = venv.emit_windows(n_step, stride)
    assert tuple(f['pixel'].shape) == (n * W, n_step, stacks * pixel[0]) + pixel[1:] and f['pixel'].dtype == torch.uint8
    px, pn = f['pixel'].cpu().numpy(), f['pixel_next'].cpu().numpy()
    for a in range(n):
        for w in range(W):
            for j in range(n_step):
                np.testing.assert_array_equal(px[a * W + w, j], host_pix[a, w * adv + j])
            np.testing.assert_array_equal(pn[a * W + w, 0], host_pix[a, w * adv + n_step])
    b = venv.to_batch(f)
    assert b['obs']['pixel']['camera0'] is f['pixel'] and b['onetime_infos'] is None
    return venv


def check_frame_stack_with_resets_inside(K, device):
    """the gather with an episode that restarts INSIDE the stored rows (episode_first): the history of a row never
    reaches back across the reset"""
    rs = np.random.RandomState(5)
    actors, R, fb, ns = 3, 9, 40, 4
    frames = rs.randint(0, 256, size=(actors, R, 2, 4, 5)).astype(np.uint8)
    first = np.zeros((actors, R), np.int32)
    first[1, 4:] = 4                    # actor 1 was reset at row 4
    first[2, 2:6] = 2
    first[2, 6:] = 6                    # actor 2 at rows 2 and 6
    start, n_step, stride, W = 1, 3, 2, 3
    dst = torch.zeros(actors * W, n_step, ns * 2, 4, 5, dtype=torch.uint8, device=device)
    K.frame_stack(torch.as_tensor(frames).to(device), ns, start, n_step, stride, W, dst,
                  episode_first=torch.as_tensor(first).to(device))
    got = dst.cpu().numpy()
    for a in range(actors):
        for w in range(W):
            for j in range(n_step):
                s = start + w * stride + j
                want = np.concatenate([frames[a, f] for f in stack_sources(s, ns, first[a, s])], 0)
                np.testing.assert_array_equal(got[a * W + w, j], want, err_msg='actor %d row %d' % (a, s))
