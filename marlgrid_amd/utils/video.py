"""Frame recorder around a batched env — the caller-side tooling of `marlgrid/utils/video.py`
(`GridRecorder`, `export_video`, `render_frames`), not on the step path.

A recorder watches ONE env of the batch (`env_index`): while it is recording, each reset()/step()
renders that env's whole-grid frame (`env.render(env_ids=[i])`, a device kernel) and keeps it in a
preallocated host buffer; the buffer is written as numbered PNGs (PIL) or as a video (moviepy, an
optional dependency upstream as well).
"""
import os

import numpy as np


def _abs(path):
    return os.path.abspath(os.path.expanduser(path))


def _as_uint8_stack(frames):
    x = np.stack(frames) if isinstance(frames, (list, tuple)) else np.asarray(frames)
    if x.dtype.kind == "f":                      # float frames are taken to be in [0, 1]
        x = np.clip(x * 255.0, 0, 255)
    return x.astype(np.uint8, copy=False)


def export_video(X, outfile, fps=30, rescale_factor=2):
    """(T, H, W, 3) frames -> a video file, each pixel blown up `rescale_factor` times (video.py:8-36)."""
    try:
        from moviepy.editor import VideoClip
    except ImportError as e:                                       # pragma: no cover
        raise ImportError("export_video needs the optional dependency moviepy") from e
    x = _as_uint8_stack(X)
    if rescale_factor not in (None, 1):
        k = int(rescale_factor)
        x = x.repeat(k, axis=1).repeat(k, axis=2)
    target = _abs(outfile)
    os.makedirs(os.path.dirname(target), exist_ok=True)
    last = len(x) - 1
    VideoClip(lambda t: x[min(int(t * fps), last)], duration=len(x) / fps).write_videofile(target, fps=fps)


def render_frames(X, path, ext="png"):
    """(T, H, W, 3) frames -> `path`/frame_<t>.<ext> (video.py:39-52)."""
    from PIL import Image
    folder = _abs(path)
    os.makedirs(folder, exist_ok=True)
    for t, frame in enumerate(_as_uint8_stack(X)):
        Image.fromarray(frame, "RGB").save(os.path.join(folder, "frame_%d.%s" % (t, ext)))


class GridRecorder(object):
    """`GridRecorder(env, save_root, ...)` behaves like the env it wraps (attribute access falls
    through) and adds the recording controls of video.py:55-178: set `recording`, or pass
    `auto_save_interval` to record every so many episodes; `export_frames()`, `export_video()`,
    `export_both()` write what has been buffered since the last reset."""

    fix_path = staticmethod(_abs)

    def __init__(self, env, save_root, max_steps=1000, auto_save_images=True, auto_save_videos=True,
                 auto_save_interval=None, render_kwargs={}, video_scale=4, env_index=0):
        self.env = env
        self.save_root = _abs(save_root)
        self.max_steps, self.env_index = int(max_steps), int(env_index)
        self.auto_save_images, self.auto_save_videos = auto_save_images, auto_save_videos
        self.auto_save_interval = auto_save_interval
        self.render_kwargs, self.video_scale = dict(render_kwargs), video_scale
        self.recording = False
        self.frames, self.ptr = None, 0          # host buffer (max_steps, H, W, 3) and its fill level
        self.reset_count, self.last_save = 0, -10000
        self.n_parallel = 1

    def __getattr__(self, name):                 # only reached for names the recorder does not define
        if name == "env" or name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def should_record(self):
        due = self.auto_save_interval is not None and self.reset_count - self.last_save >= self.auto_save_interval
        return bool(self.recording or due)

    # ---- buffering ------------------------------------------------------------------------------
    def _frame(self):
        return self.env.render(env_ids=[self.env_index], **self.render_kwargs)[0].cpu().numpy()

    def append_current_frame(self):
        if not self.should_record:
            return
        frame = self._frame()
        if self.frames is None:
            self.frames = np.empty((self.max_steps,) + frame.shape, frame.dtype)
        if self.ptr < self.max_steps:
            self.frames[self.ptr] = frame
            self.ptr += 1

    def _recorded(self):
        return self.frames[:self.ptr]

    # ---- writing ----------------------------------------------------------------------------------
    def _target(self, save_root, name):
        return os.path.join(_abs(self.save_root if save_root is None else save_root), name)

    def export_frames(self, episode_id=None, save_root=None):
        if not self.ptr:
            return None
        folder = self._target(save_root, episode_id or "frames_%d" % self.reset_count)
        render_frames(self._recorded(), folder)
        return folder

    def export_video(self, episode_id=None, save_root=None):
        if not self.ptr:
            return None
        export_video(self._recorded(), self._target(save_root, episode_id or "video_%d.mp4" % self.reset_count),
                     rescale_factor=self.video_scale)

    def export_both(self, episode_id, save_root=None):
        self.export_frames(episode_id, save_root)
        self.export_video(episode_id + ".mp4", save_root)

    # ---- env protocol -----------------------------------------------------------------------------
    def reset(self, **kwargs):
        if self.ptr and self.should_record:      # an episode was recorded: flush it before starting over
            if self.auto_save_images:
                self.export_frames()
            self.last_save = self.reset_count
        self.ptr = 0
        self.reset_count += 1
        obs = self.env.reset(**kwargs)
        self.append_current_frame()
        return obs

    def step(self, action):
        result = self.env.step(action)
        self.append_current_frame()
        return result
