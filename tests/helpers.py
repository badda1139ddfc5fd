"""Shared test helpers: fixture loading and oracle state plumbing."""
import json
import os

import numpy as np

from oracle import fairnav_oracle as fnv
from oracle import formation_oracle as fo
from oracle import nav_oracle as no

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def cfg_of(fx):
    args = json.loads(str(fx['args']))
    return no.Config(**{k: v for k, v in args.items() if k in no.Config.__dataclass_fields__})


def state_from(fx, cfg, prefix='init_'):
    n = fx[prefix + 'agent_pos'].shape[0]
    st = no.State(cfg, n)
    for k in no.State.FIELDS:
        getattr(st, k)[...] = fx[prefix + k]
    return st


TRAJ = ['traj_n3.npz', 'traj_n3w2.npz', 'traj_n1.npz', 'traj_n2o1w1.npz', 'traj_n10.npz', 'traj_n32.npz',
        'traj_n3_ep5.npz', 'traj_n3_float.npz', 'traj_n4_knobs.npz', 'traj_crafted.npz', 'traj_n3_global.npz',
        'traj_n10_global.npz']


FORM = ['form_n3.npz', 'form_n10.npz', 'form_n5l3.npz', 'form_n4_thr04.npz', 'form_n3_thr07.npz', 'form_n1.npz',
        'form_crafted.npz']


def form_cfg_of(fx):
    args = json.loads(str(fx['args']))
    kw = {k: v for k, v in args.items() if k in fo.Config.__dataclass_fields__}
    kw['num_walls'] = 2
    return fo.Config(**kw)


def form_state_from(fx, cfg, prefix='init_'):
    n = fx[prefix + 'agent_pos'].shape[0]
    st = fo.State(cfg, n)
    for k in fo.State.FIELDS:
        getattr(st, k)[...] = fx[prefix + k]
    return st


FNAV = ['fnav_n3.npz', 'fnav_n10.npz', 'fnav_n4w2.npz', 'fnav_n7_thr035.npz', 'fnav_n3_thr04.npz', 'fnav_n2.npz']


def fnav_cfg_of(fx):
    args = json.loads(str(fx['args']))
    return fnv.Config(**{k: v for k, v in args.items() if k in fnv.Config.__dataclass_fields__})


def fnav_state_from(fx, cfg, prefix='init_'):
    n = fx[prefix + 'agent_pos'].shape[0]
    st = fnv.State(cfg, n)
    for k in fnv.State.FIELDS:
        getattr(st, k)[...] = fx[prefix + k]
    return st


RUNNER = ['runner_nav.npz', 'runner_navw.npz', 'runner_form.npz', 'runner_fnav.npz', 'runner_nav10.npz', 'runner_nav32.npz', 'runner_form10.npz',
          'runner_fnav6.npz']


def runner_oracle_env(fx):
    """Oracle vec env of a runner_*.npz fixture, resets on the fixture's Philox stream (seed, env, episode)."""
    from oracle.philox import PhiloxStream
    args = json.loads(str(fx['args']))
    seed, n = int(fx['seed']), fx['obs'].shape[1]
    streams = lambda e, ep: PhiloxStream(seed, e, ep)  # noqa: E731
    sc = args['scenario_name']
    if sc == 'fair_graph_formation':
        cfg = form_cfg_of(fx)
        return fo.OracleFormationVecEnv(cfg, n, mode='dummy', streams=streams), cfg, fo.INFO_KEYS
    if sc == 'nav_fairassign_fairrew_formation_graph':
        cfg = fnav_cfg_of(fx)
        return fnv.OracleFairNavVecEnv(cfg, n, mode='dummy', streams=streams), cfg, fnv.INFO_KEYS
    cfg = cfg_of(fx)
    return no.OracleGraphVecEnv(cfg, n, mode='dummy', streams=streams), cfg, no.INFO_KEYS
