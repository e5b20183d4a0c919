"""Heightfield stand-in for the reference's ``Terrain``
(/root/reference/omniisaacgymenvs/tasks/utils/anymal_terrain_generator.py:37-68).

The generator itself is one-time CPU work and out of scope; what the hot path
depends on is its OUTPUT CONTRACT, which this class keeps:
  heightsamples  int16 (tot_rows, tot_cols) = (numLevels*80 + 400, numTerrains*80 + 400)   (:53-60)
  env_origins    float (numLevels, numTerrains, 3): patch centre, z = max height of the 2 m x 2 m
                 centre window * vertical_scale                                            (:164-171)
  horizontal_scale 0.1, vertical_scale 0.005, border_size 20, env_length / env_width, env_rows / env_cols.
Content: a curriculum of pyramid slopes, rough slopes and pyramid stairs whose difficulty grows with the
level, written from the formulas' intent, not from the reference's sub-terrain code (no scipy).
"""
from __future__ import annotations

import numpy as np


class SyntheticTerrain:
    def __init__(self, cfg, num_robots: int, seed: int = 1234) -> None:
        self.horizontal_scale = 0.1
        self.vertical_scale = 0.005
        self.border_size = 20
        self.env_length = cfg["mapLength"]
        self.env_width = cfg["mapWidth"]
        self.env_rows = cfg["numLevels"]
        self.env_cols = cfg["numTerrains"]
        self.num_maps = self.env_rows * self.env_cols
        self.env_origins = np.zeros((self.env_rows, self.env_cols, 3))
        self.width_per_env_pixels = int(self.env_width / self.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / self.horizontal_scale)
        self.border = int(self.border_size / self.horizontal_scale)
        self.tot_cols = int(self.env_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(self.env_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        rng = np.random.default_rng(seed)
        L, W = self.length_per_env_pixels, self.width_per_env_pixels
        # distance (in pixels) to the patch centre, Chebyshev: pyramids
        ii, jj = np.meshgrid(np.arange(L), np.arange(W), indexing="ij")
        cheb = np.maximum(np.abs(ii - (L - 1) / 2.0), np.abs(jj - (W - 1) / 2.0))
        platform = int(1.5 / self.horizontal_scale)  # 3 m flat platform
        for j in range(self.env_cols):
            kind = j % 3
            for i in range(self.env_rows):
                difficulty = i / self.env_rows
                rise = np.clip((L / 2.0 - cheb), 0, L / 2.0 - platform)  # 0 at the rim, max on the platform
                if kind == 0:  # pyramid slope
                    slope = 0.4 * difficulty * (1 if j % 2 == 0 else -1)
                    h = slope * rise * self.horizontal_scale
                elif kind == 1:  # rough slope
                    h = 0.4 * difficulty * rise * self.horizontal_scale
                    coarse = rng.uniform(-0.1, 0.1, size=(L // 2 + 1, W // 2 + 1))
                    h = h + np.kron(coarse, np.ones((2, 2)))[:L, :W]
                else:  # pyramid stairs
                    step_h = (0.05 + 0.175 * difficulty) * (1 if j % 2 == 0 else -1)
                    h = step_h * np.floor(rise / 3.0)
                patch = np.rint(h / self.vertical_scale).astype(np.int16)
                sx, sy = self.border + i * L, self.border + j * W
                self.height_field_raw[sx:sx + L, sy:sy + W] = patch
                x1 = int((self.env_length / 2.0 - 1) / self.horizontal_scale)
                x2 = int((self.env_length / 2.0 + 1) / self.horizontal_scale)
                y1 = int((self.env_width / 2.0 - 1) / self.horizontal_scale)
                y2 = int((self.env_width / 2.0 + 1) / self.horizontal_scale)
                self.env_origins[i, j] = [(i + 0.5) * self.env_length, (j + 0.5) * self.env_width,
                                          np.max(patch[x1:x2, y1:y2]) * self.vertical_scale]
        self.heightsamples = self.height_field_raw
