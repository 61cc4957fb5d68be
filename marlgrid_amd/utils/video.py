"""Frame recorder around a batched env — the caller-side tooling of `marlgrid/utils/video.py`
(`GridRecorder`, `export_video`, `render_frames`), not on the step path.

A recorder watches ONE env of the batch (`env_index`): while it is recording, each step() (before
the action is applied) and the reset() that ends an episode render that env's whole-grid frame (`env.render(env_ids=[i])`, a device kernel) and keeps it in a
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
    through) and adds the recording controls of video.py:55-178, with the same constructor arguments
    and defaults (`video_kwargs` merged over fps 20 / rescale_factor 1): set `recording`, or pass
    `auto_save_interval` to record every so many episodes.  As upstream, the frame of the state an
    action is taken in is captured BEFORE each step, and reset() appends the final frame and flushes
    the finished episode (`auto_save_images` -> `frames_<n>/`, `auto_save_videos` -> `video_<n>.mp4`,
    which needs moviepy — an optional dependency upstream as well).  `env_index` (new): which env of
    the batch is watched."""

    default_max_len = 1000
    default_video_kwargs = {"fps": 20, "rescale_factor": 1}
    fix_path = staticmethod(_abs)

    def __init__(self, env, save_root, max_steps=1000, auto_save_images=True, auto_save_videos=True,
                 auto_save_interval=None, render_kwargs={}, video_kwargs={}, env_index=0):
        self.env = env
        self.save_root = _abs(save_root)
        if max_steps is None:       # video.py:88-94
            ms = getattr(env, "max_steps", 0)
            max_steps = ms if ms else self.default_max_len
        self.max_steps, self.env_index = int(max_steps) + 1, int(env_index)
        self.auto_save_images, self.auto_save_videos = auto_save_images, auto_save_videos
        self.auto_save_interval = auto_save_interval
        self.render_kwargs = dict(render_kwargs)
        self.video_kwargs = {**self.default_video_kwargs, **video_kwargs}
        self.recording = False
        self.frames, self.ptr = None, 0          # host buffer (max_steps + 1, H, W, 3) and its fill level
        self.reset_count, self.last_save = 0, -10000
        self.n_parallel = getattr(env, "num_envs", 1)

    def __getattr__(self, name):                 # only reached for names the recorder does not define
        if name == "env" or name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def should_record(self):
        if self.recording:
            return True
        if self.auto_save_interval is None:
            return False
        return (self.reset_count - self.last_save) >= self.auto_save_interval

    # ---- buffering ------------------------------------------------------------------------------
    def append_current_frame(self):
        if not self.should_record:
            return
        frame = self.env.render(mode="rgb_array", env_ids=[self.env_index], **self.render_kwargs)[0].cpu().numpy()
        if self.frames is None:
            self.frames = np.zeros((self.max_steps,) + frame.shape, frame.dtype)
        if self.ptr >= self.max_steps:
            raise IndexError("GridRecorder: more than max_steps + 1 = %d frames in one episode" % self.max_steps)
        self.frames[self.ptr] = frame
        self.ptr += 1

    # ---- writing ----------------------------------------------------------------------------------
    def _target(self, save_root, name):
        return os.path.join(_abs(self.save_root if save_root is None else save_root), name)

    def export_frames(self, episode_id=None, save_root=None):
        folder = self._target(save_root, episode_id or "frames_%d" % self.reset_count)
        render_frames(self.frames[:self.ptr], folder)
        return folder

    def export_video(self, episode_id=None, save_root=None):
        export_video(self.frames[:self.ptr], self._target(save_root, episode_id or "video_%d.mp4" % self.reset_count),
                     **self.video_kwargs)

    def export_both(self, episode_id, save_root=None):
        self.export_frames("%s_frames" % episode_id, save_root=save_root)
        self.export_video("%s.mp4" % episode_id, save_root=save_root)

    # ---- env protocol (video.py:128-162) -----------------------------------------------------------
    def reset(self, **kwargs):
        if self.should_record and self.ptr > 0:
            self.append_current_frame()          # the episode's final state
            if self.auto_save_images:
                self.export_frames()
            if self.auto_save_videos:
                self.export_video()
            self.last_save = self.reset_count
        self.frames = None
        self.ptr = 0
        self.reset_count += self.n_parallel
        return self.env.reset(**kwargs)

    def step(self, action):
        self.append_current_frame()              # the state the action is taken in
        return self.env.step(action)
