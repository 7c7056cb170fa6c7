"""Host -> device input pipeline.

The reference prepares every sample on CPU workers (datasets/p3_coco.py:340-436 -> albumentations D4 + Normalize + ToTensorV2,
datasets/build_datasets.py:53-75; apply_d4_augmentations_to_lidar, p3_coco.py:115-164; collate_fn_pix2poly,
datasets/collate_funcs.py:68-116) and ships fp32 NCHW images.  Here the host only decodes files and packs bytes:

  host   : uint8 HWC tiles and the untransformed jagged point list go into one of two PINNED staging sets (a quarter of the image bytes
           of the fp32 path over PCIe), the D4 element of every tile is drawn on the host (it also has to move the vertex
           coordinates the tokenizer sees: `d4_keypoints`, the same pixel permutation as the image)
  device : on a copy stream, overlapped with the previous step's compute: H2D, then `p3_image_prepare` (D4 + Normalize + HWC->CHW in
           one pass) and `p3_points_d4` (the reference's fp32 point arithmetic); an event hands the batch to the compute stream.

No CPU fallback: the kernels come from libp3hip.so, `prepare_images` / `d4_points_` raise without it.
"""
import numpy as np
import torch

from . import hip

D4_ELEMENTS = ("e", "r90", "r180", "r270", "v", "hvt", "h", "t")      # albumentations' D4 group, index = the kernels' group id


def normalize_constants(mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), max_pixel_value=255.0):
    """albumentations.Normalize's fp32 constants: sub = mean * max_pixel_value, mul = reciprocal(std * max_pixel_value)."""
    m = np.array(mean, dtype=np.float32) * np.float32(max_pixel_value)
    s = np.array(std, dtype=np.float32) * np.float32(max_pixel_value)
    return m, np.reciprocal(s, dtype=np.float32)


def prepare_images(images_u8, groups=None, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), max_pixel_value=255.0, out=None):
    """uint8 [B,H,W,C] on the device -> fp32 [B,C,H,W] = ToTensorV2(Normalize(D4(img))) (build_datasets.py:55-75 with the reference's
    encoder config: mean 0, std 1, max 255).  groups: int32 [B] device tensor of D4 element ids, None = no augmentation."""
    C = images_u8.shape[-1]
    mean, std = (tuple(mean) + (0.0,) * C)[:C], (tuple(std) + (1.0,) * C)[:C]      # channels beyond the given constants: mean 0, std 1
    sub, mul = normalize_constants(mean, std, max_pixel_value)
    return hip.image_prepare(images_u8, groups, sub.tolist(), mul.tolist(), out=out)


def d4_points_(values, offsets, groups, in_width=224, in_height=224):
    """apply_d4_augmentations_to_lidar (p3_coco.py:115-164) on the whole jagged batch, in place, on the device."""
    return hip.points_d4_(values, offsets, groups, float(in_width // 2), float(in_height // 2))


def prepare_ffl_targets(gt_polygons_u8=None, crossfield_angle_u8=None, distances=None, sizes=None, groups=None):
    """FFL ground truth of a batch on the device (datasets/p3_coco.py:254-296): uint8 [B,H,W,3] polygon masks -> fp32 [B,3,H,W] in
    [0, 1]; uint8 [B,H,W] crossfield angle -> radians, normals -> tangents, rotated / mirrored with the tile; `distances` / `sizes`
    fp32 [B,H,W] -> [B,1,H,W]; all through the tile's D4 permutation (groups int32 [B], None = no augmentation).
    Returns the gt_batch entries the FFL criterion reads."""
    return hip.ffl_targets_prepare(gt_polygons_u8, crossfield_angle_u8, distances, sizes, groups)


def d4_keypoints(coords_yx, element, height, width):
    """Where D4 `element` moves integer pixel coordinates (y, x) - the image permutation of `p3_image_prepare` applied to vertex
    coordinates (albumentations does this for the `keypoints=` of p3_coco.py:424 in 'yx' format).  Host side, numpy: <= 192 vertices."""
    c = np.asarray(coords_yx)
    if c.size == 0:
        return c.copy()
    y, x = c[..., 0], c[..., 1]
    n_y, n_x = height - 1, width - 1
    g = D4_ELEMENTS.index(element) if isinstance(element, str) else int(element)
    ny, nx = {0: (y, x), 1: (n_x - x, y), 2: (n_y - y, n_x - x), 3: (x, n_y - y), 4: (n_y - y, x), 5: (n_x - x, n_y - y), 6: (y, n_x - x),
              7: (x, y)}[g]
    return np.stack([ny, nx], axis=-1)


def pack_lidar(clouds, values_out=None, offsets_out=None):
    """list of [n_i, 3] float32 arrays / tensors -> (values [sum n_i, 3], offsets int64 [B+1]) - the (values, offsets) pair of the
    nested jagged tensor collate_fn_pix2poly builds (collate_funcs.py:110-112); writes into the given (pinned) buffers when passed."""
    lens = [int(c.shape[0]) for c in clouds]
    total = sum(lens)
    offsets = torch.zeros(len(clouds) + 1, dtype=torch.int64) if offsets_out is None else offsets_out[: len(clouds) + 1]
    offsets[0] = 0
    offsets[1:] = torch.tensor(lens, dtype=torch.int64).cumsum(0)
    if values_out is None:
        values = torch.empty((total, 3), dtype=torch.float32)
    else:
        if values_out.shape[0] < total:
            raise hip.P3Error(f"pack_lidar: staging buffer holds {values_out.shape[0]} points, batch has {total}")
        values = values_out[:total]
    o = 0
    for c, n in zip(clouds, lens):
        values[o:o + n].copy_(torch.as_tensor(c, dtype=torch.float32))
        o += n
    return values, offsets


def _stop_feeder(stop, free):
    stop.set()
    free.put(None)                   # wakes a feeder parked on free.get()


def _feeder_loop(ref, it, free, ready, stop, device):
    try:
        torch.cuda.set_device(device)
        for host in it:
            s = free.get()
            if stop.is_set() or s is None:
                return
            pf = ref()
            if pf is None:
                return
            if s["copied"] is not None:
                s["copied"].synchronize()                  # the H2D copies that last read this set's pinned buffers are done
            out, rdy = pf._upload(s, pf._stage(s, host))
            del pf
            ready.put((s, out, rdy, None))
        ready.put((None, None, None, StopIteration()))
    except BaseException as e:                             # surface feeder errors in the consumer's thread
        ready.put((None, None, None, e))


class DevicePrefetcher:
    """Triple-buffered host->device feeder.  Wraps an iterable of HOST batches (dicts):
         "image"  uint8 [B,H,W,C] tensor / array (optional)       "lidar"  list of [n_i,3] float32 clouds (optional)
         "group"  D4 element ids, int [B] (optional: augmentation)  anything else: tensors copied as they are (tokens, perm matrices)
         "gt_polygons_image" uint8 [B,H,W,3] (+ "gt_crossfield_angle" uint8 [B,H,W], "distances" / "sizes" float [B,H,W]): FFL ground
                  truth as stored on disk -> the fp32 NCHW gt_batch entries of the FFL criterion (`prepare_ffl_targets`)
       and yields DEVICE batches {"image": fp32 [B,C,H,W], "lidar_values", "lidar_offsets", ...} ready for the model.

    A feeder thread packs batch k+1 / k+2 into pinned staging and queues the H2D copies + prepare kernels on `self.stream` while the
    caller's thread launches step k, so neither the host packing nor the copies sit on the step's critical path.  Every staging
    set owns its pinned buffers AND its device buffers for the life of the feeder (shape-stable batches allocate nothing after the
    first `depth` batches; round 1 allocated ~60 MB of device tensors per step on the side stream, which on a fresh box fell through
    to hipMalloc inside the timed loop).  Hand-over is by events: `ready` (copy stream -> consumer stream), `consumed` (consumer stream
    -> copy stream, recorded when the caller asks for the next batch).  A yielded batch stays valid until the next call of next();
    work already enqueued on the consumer's stream at that moment is safe."""

    def __init__(self, batches, device, max_points=1 << 20, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), max_pixel_value=255.0,
                 in_width=224, in_height=224, depth=3):
        import queue
        import threading
        self.it = iter(batches)
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.Stream(device=self.device)
        self.norm = (mean, std, max_pixel_value)
        self.size = (in_width, in_height)
        self.max_points = max_points
        self.free, self.ready = queue.Queue(), queue.Queue()
        for _ in range(max(2, depth)):
            self.free.put({"pin": {}, "dev": {}, "copied": None, "consumed": None})
        self.current = None
        # The feeder thread must not keep the prefetcher alive (with a bound method as its target an abandoned iterator was never
        # collected, so nothing ever stopped the thread or released the pinned / device staging).  It holds a weak reference, takes a strong one
        # only while it packs a batch, and a finalizer - run by close(), by garbage collection or at interpreter exit - wakes and stops it.
        import weakref
        self._stop = threading.Event()
        self._finalizer = weakref.finalize(self, _stop_feeder, self._stop, self.free)
        self.thread = threading.Thread(target=_feeder_loop, args=(weakref.ref(self), self.it, self.free, self.ready, self._stop, self.device),
                                       name="p3-feeder", daemon=True)
        self.thread.start()

    # ---- persistent buffers of one staging set
    @staticmethod
    def _buf(store, key, shape, dtype, make):
        buf = store.get(key)
        if buf is None or buf.dtype != dtype or buf.dim() != len(shape) or any(a < b for a, b in zip(buf.shape, shape)):
            buf = make(shape, dtype)
            store[key] = buf
        return buf[tuple(slice(0, n) for n in shape)]

    def _pinned(self, s, key, shape, dtype):
        return self._buf(s["pin"], key, shape, dtype, lambda sh, dt: torch.empty(sh, dtype=dt).pin_memory())

    def _device(self, s, key, shape, dtype):
        return self._buf(s["dev"], key, shape, dtype, lambda sh, dt: torch.empty(sh, dtype=dt, device=self.device))

    def _stage(self, s, host):
        """host batch -> pinned staging of set `s` (CPU work of the feeder thread)"""
        staged = {}
        if host.get("image") is not None:
            img = torch.as_tensor(host["image"])
            staged["image"] = self._pinned(s, "image", tuple(img.shape), torch.uint8)
            staged["image"].copy_(img)
        if host.get("lidar") is not None:
            vals = self._pinned(s, "lidar_values", (self.max_points, 3), torch.float32)
            offs = self._pinned(s, "lidar_offsets", (len(host["lidar"]) + 1,), torch.int64)
            staged["lidar_values"], staged["lidar_offsets"] = pack_lidar(host["lidar"], vals, offs)
        if host.get("group") is not None:
            g = torch.as_tensor(host["group"], dtype=torch.int32)
            staged["group"] = self._pinned(s, "group", tuple(g.shape), torch.int32)
            staged["group"].copy_(g)
        for k, v in host.items():
            if k in ("image", "lidar", "group") or v is None:
                continue
            t = torch.as_tensor(v)
            staged[k] = self._pinned(s, k, tuple(t.shape), t.dtype)
            staged[k].copy_(t)
        return staged

    def _upload(self, s, staged):
        """pinned staging -> device buffers of set `s` + prepare kernels, all on the copy stream; returns the device batch"""
        out = {}
        with torch.cuda.stream(self.stream):
            if s["consumed"] is not None:
                self.stream.wait_event(s["consumed"])          # the consumer's kernels that read this set's last batch are done
            dev = {}
            for k, v in staged.items():
                d = self._device(s, "raw_" + k, tuple(v.shape), v.dtype)
                d.copy_(v, non_blocking=True)
                dev[k] = d
            s["copied"] = torch.cuda.Event()
            s["copied"].record(self.stream)
            grp = dev.get("group")
            if "image" in dev:
                B, H, W, C = dev["image"].shape
                out["image"] = prepare_images(dev["image"], grp, *self.norm, out=self._device(s, "image", (B, C, H, W), torch.float32))
            if "lidar_values" in dev:
                out["lidar_values"], out["lidar_offsets"] = dev["lidar_values"], dev["lidar_offsets"]
                if grp is not None:
                    d4_points_(out["lidar_values"], out["lidar_offsets"], grp, *self.size)
            if "gt_polygons_image" in dev and dev["gt_polygons_image"].dtype == torch.uint8:       # FFL ground truth as stored (uint8 masks)
                out.update(prepare_ffl_targets(dev["gt_polygons_image"], dev.get("gt_crossfield_angle"), dev.get("distances"), dev.get("sizes"), grp))
            for k, v in dev.items():
                if k not in ("image", "lidar_values", "lidar_offsets", "group") and k not in out:
                    out[k] = v
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return out, ready

    def close(self):
        """stop the feeder thread (it may be parked on free.get()) and drop the pinned / device staging; idempotent"""
        self._finalizer()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __iter__(self):
        return self

    def __next__(self):
        cur = torch.cuda.current_stream(self.device)
        if self.current is not None:                           # hand the previous set back: its consumers are enqueued by now
            ev = torch.cuda.Event()
            ev.record(cur)
            self.current["consumed"] = ev
            self.free.put(self.current)
            self.current = None
        s, out, ready, err = self.ready.get()
        if err is not None:
            self.ready.put((None, None, None, err))            # stay exhausted / keep raising
            if isinstance(err, StopIteration):
                self.close()
                raise StopIteration
            raise err
        cur.wait_event(ready)
        self.current = s
        return out
