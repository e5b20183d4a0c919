"""CPU: the oracle reproduces the golden vectors of the in-hand family (ShadowHand, AllegroHand), Anymal (flat) and
AnymalTerrain (tests/golden/*.npz, produced by oracle/gen_golden.py running the reference's own method source on stub
objects).  Runs without /root/reference: the fixtures hold every input, every output and the random draws."""
import os

import pytest
import torch

import golden_io
from helpers import assert_close, assert_equal
from oracle import anymal_oracle as an
from oracle import anymal_terrain_oracle as ao
from oracle import hand_oracle as ho
from test_oracle_pinned_hand import SPECS as HAND_SPECS

HAND_KINDS = ["shadow_openai_ff", "shadow_full", "allegro_full", "allegro_mcs_ema"]


def hand_case(kind, path):
    g = golden_io.load(path)
    spec = HAND_SPECS[kind]()
    s = golden_io.unflat(g, "s", ho.HandState)
    b = golden_io.unflat(g, "b", ho.HandBuffers)
    init = golden_io.unflat(g, "init", ho.HandInit)
    return g, spec, s, b, init


@pytest.mark.parametrize("kind", HAND_KINDS)
def test_hand_oracle_reproduces_golden(kind):
    paths = golden_io.files(f"hand_{kind}")
    assert len(paths) == 2
    for path in paths:
        tag = os.path.basename(path)
        g, spec, s, b, init = hand_case(kind, path)
        post = b.clone()
        ho.post_physics(spec, s, post)
        names = ["obs_buf", "rew_buf", "reset_buf", "reset_goal_buf", "progress_buf", "successes", "consecutive_successes"]
        if spec.asymmetric_obs:
            names.append("states_buf")
        for k in names:
            assert torch.equal(getattr(post, k), g[f"post_{k}"]), f"{tag} post {k}"
        assert int(post.reset_buf.sum()) > 0 and int(post.reset_goal_buf.sum()) > 0
        pre = b.clone()
        targets, goal_ids, goal_rows, env_ids, rows = ho.pre_physics(
            spec, pre, init, g["pre_actions"], draw_goal=lambda ids: g["pre_u_goal"],
            draw_reset=lambda ids: (g["pre_u_reset"], g["pre_u_reset_goal"]))
        for k in ("reset_buf", "reset_goal_buf", "progress_buf", "successes", "goal_pos", "goal_rot", "actions", "prev_targets",
                  "cur_targets"):
            assert torch.equal(getattr(pre, k), g[f"pre_{k}"]), f"{tag} pre {k}"
        assert torch.equal(targets, g["pre_targets"])
        assert torch.equal(rows["dof_pos"], g["pre_dof_pos"]) and torch.equal(rows["dof_vel"], g["pre_dof_vel"])
        assert torch.equal(rows["object_pos"], g["pre_object_pos"]) and torch.equal(rows["object_rot"], g["pre_object_rot"])
        assert torch.equal(goal_rows[0], g["pre_goal_world_pos"]) and torch.equal(goal_rows[1], g["pre_goal_rot_rows"])
        assert torch.equal(rows["goal_world_pos"], g["pre_r_goal_world_pos"]) and torch.equal(rows["goal_rot"], g["pre_r_goal_rot"])


def anymal_flat_case(path):
    g = golden_io.load(path)
    st = {k[3:]: v for k, v in g.items() if k.startswith("st_")}
    b = golden_io.unflat(g, "b", an.AnymalFlatBuffers)
    return g, an.AnymalFlatSpec(), st, b


def test_anymal_flat_oracle_reproduces_golden():
    paths = golden_io.files("anymalflat")
    assert len(paths) == 2
    for path in paths:
        tag = os.path.basename(path)
        g, spec, st, b = anymal_flat_case(path)
        post = b.clone()
        an.post_physics(spec, post, st["pos"], st["rot"], st["vel"], st["dof_pos"], st["dof_vel"])
        for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "last_actions", "last_dof_vel"):
            assert torch.equal(getattr(post, k), g[f"post_{k}"]), f"{tag} post {k}"
        pre = b.clone()
        rows = an.pre_physics(spec, pre, g["pre_actions"], g["lower"], g["upper"], g["pre_initial_root_pos"],
                              g["pre_initial_root_rot"], draw=lambda ids: tuple(g[f"pre_u{i}"] for i in range(4)))
        for k in ("reset_buf", "progress_buf", "commands", "actions", "last_actions", "last_dof_vel", "current_targets"):
            assert torch.equal(getattr(pre, k), g[f"pre_{k}"]), f"{tag} pre {k}"
        assert torch.equal(rows.dof_pos, g["pre_dof_pos"]) and torch.equal(rows.dof_vel, g["pre_dof_vel"])
        assert torch.equal(rows.root_pos, g["pre_root_pos"]) and torch.equal(rows.root_rot, g["pre_root_rot"])
        assert torch.equal(rows.root_vel, g["pre_root_vel"])


def anymal_terrain_case(path):
    g = golden_io.load(path)
    seed = int(g["hs_seed"])
    hs = torch.randint(-60, 200, (1200, 2000), generator=torch.Generator().manual_seed(seed), dtype=torch.int16)
    assert int(hs.long().sum()) == int(g["hs_sum"]), "heightfield regeneration drifted (torch.randint stream changed)"
    s = golden_io.unflat(g, "s", ao.AnymalState)
    b = golden_io.unflat(g, "b", ao.AnymalBuffers)
    return g, ao.AnymalSpec(), s, b, hs, g["origins"]


def test_anymal_terrain_oracle_reproduces_golden():
    paths = golden_io.files("anymalterrain")
    assert len(paths) == 2
    for path in paths:
        tag = os.path.basename(path)
        g, spec, s, b, hs, origins = anymal_terrain_case(path)
        draw = lambda ids: ao.AnymalResetDraws(*[g[f"draw{i}"] for i in range(6)])
        env_ids, view = ao.post_physics(spec, b, s, hs, origins, draw=draw, noise_u=g["noise_u"])
        assert len(env_ids) > 0
        for k in ("obs_buf", "rew_buf", "reset_buf", "timeout_buf", "progress_buf", "commands", "last_actions", "last_dof_vel",
                  "feet_air_time", "terrain_levels", "env_origins"):
            assert torch.equal(getattr(b, k), g[f"post_{k}"]), f"{tag} {k}"
        for k in ao.EPISODE_KEYS:
            assert torch.equal(b.episode_sums[k], g[f"post_episode_sums__{k}"]), k
            assert torch.equal(b.extras["episode"]["rew_" + k], g[f"post_extras__{k}"]), k
        assert torch.equal(b.extras["episode"]["terrain_level"], g["post_extras__terrain_level"])
        assert torch.equal(view["base_pos"][env_ids], g["reset_base_pos"]) and torch.equal(view["dof_pos"][env_ids], g["reset_dof_pos"])
        assert torch.equal(view["dof_vel"][env_ids], g["reset_dof_vel"]) and torch.equal(view["base_quat"][env_ids], g["reset_base_quat"])
