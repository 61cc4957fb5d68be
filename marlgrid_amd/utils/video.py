"""Frame recorder around a batched env (the reference's `marlgrid/utils/video.py`: `GridRecorder`,
`export_video`, `render_frames`).  Caller-side tooling, not on the step path: it buffers
`env.render()` frames of ONE env of the batch (`env_index`) and writes them with PIL (PNG frames)
or moviepy (video, if that optional dependency is installed — it is optional upstream as well).
"""
import os

import numpy as np


def export_video(X, outfile, fps=30, rescale_factor=2):
    """X: (T, H, W, 3) uint8 frames -> video file (needs moviepy, like upstream video.py:8-36)."""
    try:
        import moviepy.editor as mpy
    except ImportError as e:                                       # pragma: no cover
        raise ImportError("GridRecorder.export_video requires moviepy") from e
    if isinstance(X, list):
        X = np.stack(X)
    if isinstance(X, np.floating) or X.dtype.kind == "f":
        X = (X * 255).astype(np.uint8).clip(0, 255)
    if rescale_factor is not None and rescale_factor != 1:
        X = np.kron(X, np.ones((1, int(rescale_factor), int(rescale_factor), 1))).astype(np.uint8)

    def make_frame(i):
        out = X[i]
        return out
    getframe = lambda t: make_frame(min(int(t * fps), len(X) - 1))
    clip = mpy.VideoClip(getframe, duration=len(X) / fps)
    outfile = os.path.abspath(os.path.expanduser(outfile))
    os.makedirs(os.path.dirname(outfile), exist_ok=True)
    clip.write_videofile(outfile, fps=fps)


def render_frames(X, path, ext="png"):
    """Write frames as numbered images with PIL (upstream video.py:39-52)."""
    from PIL import Image
    path = os.path.abspath(os.path.expanduser(path))
    os.makedirs(path, exist_ok=True)
    if isinstance(X, list):
        X = np.stack(X)
    for k, frame in enumerate(X):
        Image.fromarray(np.asarray(frame, dtype=np.uint8), "RGB").save(os.path.join(path, "frame_%d.%s" % (k, ext)))


class GridRecorder(object):
    """Wraps a `MultiGridEnv`; while `recording` is set, every reset()/step() appends the rendered
    frame (and, in step, action / reward / done of the watched env) to buffers that
    `export_frames()` / `export_video()` write out (upstream video.py:55-178)."""

    def __init__(self, env, save_root, max_steps=1000, auto_save_images=True, auto_save_videos=True,
                 auto_save_interval=None, render_kwargs={}, video_scale=4, env_index=0):
        self.env = env
        self.frames = None
        self.ptr = 0
        self.reset_count = 0
        self.last_save = -10000
        self.recording = False
        self.save_root = self.fix_path(save_root)
        self.auto_save_videos = auto_save_videos
        self.auto_save_images = auto_save_images
        self.auto_save_interval = auto_save_interval
        self.render_kwargs = dict(render_kwargs)
        self.video_scale = video_scale
        self.env_index = int(env_index)
        self.max_steps = int(max_steps)
        self.n_parallel = 1

    def __getattr__(self, name):
        if name.startswith("_") or name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    @staticmethod
    def fix_path(path):
        return os.path.abspath(os.path.expanduser(path))

    @property
    def should_record(self):
        if self.recording:
            return True
        if self.auto_save_interval is None:
            return False
        return (self.reset_count - self.last_save) >= self.auto_save_interval

    def export_frames(self, episode_id=None, save_root=None):
        if self.ptr == 0:
            return None
        if save_root is None:
            save_root = self.save_root
        if episode_id is None:
            episode_id = "frames_%d" % self.reset_count
        path = os.path.join(self.fix_path(save_root), episode_id)
        render_frames(self.frames[:self.ptr], path)
        return path

    def export_video(self, episode_id=None, save_root=None):
        if self.ptr == 0:
            return None
        if save_root is None:
            save_root = self.save_root
        if episode_id is None:
            episode_id = "video_%d.mp4" % self.reset_count
        export_video(self.frames[:self.ptr], os.path.join(self.fix_path(save_root), episode_id),
                     rescale_factor=self.video_scale)

    def export_both(self, episode_id, save_root=None):
        self.export_frames(episode_id, save_root)
        self.export_video(episode_id + ".mp4", save_root)

    def _frame(self):
        img = self.env.render(env_ids=[self.env_index], **self.render_kwargs)
        return img[0].cpu().numpy()

    def append_current_frame(self):
        if not self.should_record:
            return
        new_frame = self._frame()
        if self.frames is None:
            self.frames = np.zeros((self.max_steps, *new_frame.shape), dtype=new_frame.dtype)
        if self.ptr < self.max_steps:
            self.frames[self.ptr] = new_frame
            self.ptr += 1

    def reset(self, **kwargs):
        if self.should_record and self.ptr > 0:
            if self.auto_save_images:
                self.export_frames()
            self.last_save = self.reset_count
        self.ptr = 0
        self.reset_count += 1
        obs = self.env.reset(**kwargs)
        self.append_current_frame()
        return obs

    def step(self, action):
        out = self.env.step(action)
        self.append_current_frame()
        return out
