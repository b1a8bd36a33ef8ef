"""SURVEY 8f row N1, the optional sink behind the orbit renderer: what the reference does with the frames it has rendered
(/root/reference/utils/visualization.py:343-362 -> create_animation, :180-201, :244-266): `(frame * 255).astype(uint8)`,
channels last, PIL images, an animated GIF.  Here the conversion runs on the GPU (one byte per channel crosses PCIe, not
four), the copy to pinned host memory runs on a side stream while the next block of views is rendered, and encoding runs
on a worker thread.  A path ending in .gif gives the reference's animated GIF (same duration / loop arguments); anything
else is taken as a directory and gets one PNG per frame.  Values are clamped to [0, 1] first: the reference's bare
astype wraps out-of-range values around, which no caller wants; in-range values give the same bytes.  MP4 needs OpenCV
(visualization.py:268-290), which this image does not have."""
import os
import queue
import threading

import numpy as np
import torch


class FrameSink:
    def __init__(self, path, fps=30):
        self.path = os.fspath(path)
        self.gif = self.path.lower().endswith(".gif")
        if self.path.lower().endswith(".mp4"):
            raise ValueError("FrameSink writes .gif or a directory of PNGs (MP4 needs OpenCV, absent here)")
        self.fps = fps
        self.count = 0
        self._frames = []                 # GIF: PIL images until close()
        self._q = queue.Queue(maxsize=4)
        self._err = None
        self._stream = None
        if not self.gif:
            os.makedirs(self.path, exist_ok=True)
        self._worker = threading.Thread(target=self._drain, daemon=True)
        self._worker.start()

    # ---- producer side ------------------------------------------------------------------------------------------
    def write(self, frames):
        """frames: (V, 3, H, W) or (3, H, W), float, on the GPU or the host.  Returns at once; the bytes are on their way."""
        if self._err is not None:
            raise self._err
        if frames.dim() == 3:
            frames = frames.unsqueeze(0)
        first = self.count
        self.count += frames.shape[0]
        if not frames.is_cuda:
            u8 = (frames.detach().float().clamp(0.0, 1.0) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            self._q.put((first, u8, None))
            return
        dev = frames.device
        if self._stream is None or self._stream.device != dev:
            self._stream = torch.cuda.Stream(device=dev)
        ready = torch.cuda.Event(); ready.record(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._stream):
            self._stream.wait_event(ready)
            u8 = (frames.detach().float().clamp(0.0, 1.0) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            frames.record_stream(self._stream)
            host = torch.empty(u8.shape, dtype=torch.uint8).pin_memory()
            host.copy_(u8, non_blocking=True)
            done = torch.cuda.Event(); done.record(self._stream)
        self._q.put((first, host, done))

    def close(self):
        self._q.put(None)
        self._worker.join()
        if self._err is not None:
            raise self._err
        if self.gif and self._frames:
            self._frames[0].save(self.path, save_all=True, append_images=self._frames[1:], duration=1000 / self.fps, loop=0)
        return self.count

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- consumer side (worker thread) ---------------------------------------------------------------------------
    def _drain(self):
        from PIL import Image
        while True:
            item = self._q.get()
            if item is None:
                return
            try:
                first, host, done = item
                if done is not None:
                    done.synchronize()
                arr = host.numpy()
                for k in range(arr.shape[0]):
                    im = Image.fromarray(np.ascontiguousarray(arr[k]))
                    if self.gif:
                        self._frames.append(im)
                    else:
                        im.save(os.path.join(self.path, f"frame_{first + k:04d}.png"))
            except Exception as e:            # surfaces at the next write() / close()
                self._err = e
