"""Binary cache of a range-only dataset, the format of slam_amd/host/slam_data.hpp
(write_dataset_cache / read_dataset_cache): a 64-byte header, then flat
little-endian arrays.

    bytes 0..7   b"SLAMRO1\\0"
    u32 version (1), u32 control dimension (3), u32 reading dimension (1), u32 0
    u64 T, u64 N, 24 zero bytes
    f64 dt[T]; f64 control[3][T] (component-major)
    u32 timestep[N]; u32 landmark id[N], each padded to a multiple of 8 bytes
    f64 range[N], arrival order

Controls are the raw (v, w, slip) rates of the log (src/dataset.cpp:65-76 in the
reference divides the increments by dt); the control model builder turns them
into means and deviations.
"""
import struct

import numpy as np

MAGIC = b"SLAMRO1\0"
_HEADER = struct.Struct("<8sIIIIQQ24s")


def write(path, dt, controls, obs_t, obs_id, obs_range):
    dt = np.ascontiguousarray(dt, dtype="<f8")
    controls = np.ascontiguousarray(controls, dtype="<f8").reshape(-1, 3)
    obs_t = np.ascontiguousarray(obs_t, dtype="<u4")
    obs_id = np.ascontiguousarray(obs_id, dtype="<u4")
    obs_range = np.ascontiguousarray(obs_range, dtype="<f8")
    T, N = dt.size, obs_t.size
    if controls.shape[0] != T or obs_id.size != N or obs_range.size != N:
        raise ValueError("array lengths disagree")
    if N and (np.any(np.diff(obs_t.astype(np.int64)) < 0) or int(obs_t.max()) > T):
        raise ValueError("readings must be in arrival order with timesteps in [0, T]")
    pad = b"\0" * (4 * (N & 1))
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, 1, 3, 1, 0, T, N, b"\0" * 24))
        f.write(dt.tobytes())
        f.write(np.ascontiguousarray(controls.T).tobytes())
        f.write(obs_t.tobytes() + pad)
        f.write(obs_id.tobytes() + pad)
        f.write(obs_range.tobytes())


def read(path):
    with open(path, "rb") as f:
        raw = f.read(_HEADER.size)
        if len(raw) != _HEADER.size:
            raise ValueError(f"{path} is truncated")
        magic, version, cdim, rdim, _, T, N, _ = _HEADER.unpack(raw)
        if magic != MAGIC or version != 1 or cdim != 3 or rdim != 1:
            raise ValueError(f"{path} is not a version-1 range-only dataset cache")
        Np = N + (N & 1)

        def take(dtype, count):
            a = np.fromfile(f, dtype=dtype, count=count)
            if a.size != count:
                raise ValueError(f"{path} is truncated")
            return a

        dt = take("<f8", T)
        controls = take("<f8", 3 * T).reshape(3, T).T.copy()
        obs_t = take("<u4", Np)[:N].copy()
        obs_id = take("<u4", Np)[:N].copy()
        obs_range = take("<f8", N)
    return {"T": T, "N": N, "dt": dt, "controls": controls, "obs_t": obs_t, "obs_id": obs_id, "obs_range": obs_range}


def dense_landmarks(obs_id):
    """Landmark ids -> (dense indices numbered by first appearance, the ids in that order): the numbering
    the C ABI takes (the index into the reference's feature_estimates vector)."""
    obs_id = np.asarray(obs_id)
    _, first = np.unique(obs_id, return_index=True)
    order = obs_id[np.sort(first)]
    remap = np.full(int(obs_id.max()) + 1 if obs_id.size else 1, -1, dtype=np.int64)
    remap[order] = np.arange(order.size)
    return remap[obs_id].astype(np.uint32), order
