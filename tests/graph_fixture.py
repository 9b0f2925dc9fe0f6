"""Synthetic dataset / encoder collaborators for driving holoagent_amd.graph.Graph in tests and smoke()."""
import numpy as np

from holoagent_amd.synth import SceneSpec, SynthScene


class SynthDataset:
    def __init__(self, scene: SynthScene):
        self.scene = scene
        self.frames = [scene.frame(i) for i in range(scene.spec.n_frames)]

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        f = self.frames[i]
        return f["rgb"], f["depth"], f["pose"], None, f["K"]

    def get_camera_intrinsics(self):
        return self.frames[0]["K"]


class SynthEncoders:
    """Stands in for SAM + CLIP: returns the frame's masks / features, and a deterministic text table."""

    def __init__(self, ds: SynthDataset, words):
        self.ds = ds
        self._by_rgb = {f["rgb"].tobytes()[:4096] + bytes([i % 251]): i for i, f in enumerate(ds.frames)}
        self._calls = 0
        rng = np.random.Generator(np.random.PCG64(2024))
        D = ds.scene.spec.feat_dim
        self.table = {}
        for w in words:
            v = rng.standard_normal(D).astype(np.float32)
            self.table[w] = v / np.linalg.norm(v)

    def extract(self, rgb):
        f = self.ds.frames[self._calls % len(self.ds.frames)]
        self._calls += 1
        return dict(masks=f["masks"], f_g=f["f_g"], f_masked=f["f_masked"], f_crop=f["f_crop"])

    def encode_text(self, prompts):
        out = []
        for p in prompts:
            key = p.replace("a photo of ", "").replace(" in the scene.", "")
            base = self.table.setdefault(key, self._new(key))
            v = base + (0.05 if p != key else 0.0) * self._new(p + "#t")
            out.append(v / np.linalg.norm(v))
        return np.stack(out).astype(np.float32)

    def _new(self, key):
        rng = np.random.Generator(np.random.PCG64(abs(hash(key)) % (2 ** 32)))
        v = rng.standard_normal(self.ds.scene.spec.feat_dim).astype(np.float32)
        return v / np.linalg.norm(v)


def tiny_scene(n_frames=8, D=32):
    spec = SceneSpec(seed=5, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=96, height=72,
                     n_frames=n_frames, n_masks=8, feat_dim=D, yaw_step_deg=40.0)
    return SynthScene(spec)
