"""Wire formats of the solver outputs: the five ``stein_msgs`` ROS 2 messages, without ROS.

SURVEY.md §8(f)-2.  The message schemas below are the field lists of the reference's interface package
(/root/reference/stein_msgs/msg/{SteinParticle,SteinParticleArray,SteinParameters,Runtime,Variance}.msg) and
``fill_*`` mirror how the odometry node fills them (OdometryPipeline.cpp:940-1020: ``SteinParticle`` is the
``[6·P]`` particle vector sliced into x, y, z, roll, pitch, yaw plus the weights).  ``encode`` / ``decode``
implement the serialization ROS 2 puts on the wire and into bags for such messages — OMG CDR, little endian,
with the 4-byte encapsulation header ``00 01 00 00``: primitives aligned to their size relative to the start of
the body, ``string`` = uint32 length (incl. NUL) + bytes + NUL, unbounded sequences = uint32 count + elements,
fixed arrays = elements only, nested messages in place.

Parity status: *unpinned* — rosidl / rmw are not in this image, so the byte streams are checked against the
CDR rules by hand-computed vectors (tests/test_pipeline_cpu.py), not against a ROS installation.
``python -m svnicp_amd.stein_msgs --emit DIR`` writes ``.msg`` definition files generated from these schemas,
for a colcon workspace that wants the interface package.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

# (field name, type).  Types: bool, int16, int64, float64, string, ("seq", T), ("arr", T, N), or a message name.
SCHEMAS: dict[str, list[tuple]] = {
    "builtin_interfaces/Time": [("sec", "int32"), ("nanosec", "uint32")],
    "std_msgs/Header": [("stamp", "builtin_interfaces/Time"), ("frame_id", "string")],
    "stein_msgs/SteinParticle": [("header", "std_msgs/Header")] + [(n, ("seq", "float64")) for n in
                                 ("x", "y", "z", "roll", "pitch", "yaw", "weights")],
    "stein_msgs/SteinParticleArray": [("header", "std_msgs/Header"), ("stein_particle_array", ("seq", "stein_msgs/SteinParticle"))],
    "stein_msgs/SteinParameters": [("header", "std_msgs/Header"), ("optimizer", "string"), ("iterations", "int64"),
                                   ("batch_size", "int64"), ("particle_count", "int64"), ("normalize", "bool"),
                                   ("learning_rate", "float64"), ("correspondence_distance", "float64"), ("early_stop", "bool"),
                                   ("converge_steps", "int16"), ("converge_threshold", "float64"), ("deskew_cloud", "bool"),
                                   ("voxelization", "bool"), ("voxel_size", "float64"), ("map_voxel_size", "float64"),
                                   ("map_voxel_max_points", "float64"), ("point_range", ("arr", "float64", 2)),
                                   ("weight_mean", "bool"), ("md_learning_rate", "float64"), ("md_iterations", "int64")],
    "stein_msgs/Runtime": [("header", "std_msgs/Header")] + [(n, "float64") for n in
                           ("steinicp_time", "preprocessing_time", "knn_time", "update_time", "finish_iter")],
    "stein_msgs/Variance": [("header", "std_msgs/Header")] + [(n, ("arr", "float64", 6)) for n in
                            ("var_icp", "var_mean_filtered", "var_maxsliding_filtered", "var_random_walk")],
}
_PRIM = {"bool": ("<?", 1), "int16": ("<h", 2), "int32": ("<i", 4), "uint32": ("<I", 4), "int64": ("<q", 8), "float64": ("<d", 8)}


@dataclass
class Msg:
    """A message instance: type name + field dict (nested messages are ``Msg`` too)."""
    type: str
    fields: dict = field(default_factory=dict)

    def __getitem__(self, k):
        return self.fields[k]


def default(type_name: str) -> Msg:
    m = Msg(type_name)
    for name, t in SCHEMAS[type_name]:
        m.fields[name] = _default_value(t)
    return m


def _default_value(t):
    if isinstance(t, tuple):
        if t[0] == "seq":
            return []
        return [_default_value(t[1]) for _ in range(t[2])]
    if t == "string":
        return ""
    if t == "bool":
        return False
    if t in _PRIM:
        return 0.0 if t == "float64" else 0
    return default(t)


# ----------------------------------------------------------------------------- CDR
class _Writer:
    def __init__(self):
        self.b = bytearray()

    def align(self, n):
        self.b.extend(b"\0" * ((-len(self.b)) % n))

    def prim(self, t, v):
        fmt, size = _PRIM[t]
        self.align(size)
        self.b.extend(struct.pack(fmt, v))

    def value(self, t, v):
        if isinstance(t, tuple):
            if t[0] == "seq":
                self.prim("uint32", len(v))
                if t[1] == "float64" and len(v):
                    self.align(8)
                    self.b.extend(np.asarray(v, "<f8").tobytes())
                else:
                    for e in v:
                        self.value(t[1], e)
            else:
                if len(v) != t[2]:
                    raise ValueError(f"fixed array of {t[2]} expected, got {len(v)}")
                for e in v:
                    self.value(t[1], e)
        elif t == "string":
            raw = v.encode("utf-8") + b"\0"
            self.prim("uint32", len(raw))
            self.b.extend(raw)
        elif t in _PRIM:
            self.prim(t, v)
        else:
            for name, ft in SCHEMAS[t]:
                self.value(ft, v.fields[name])


class _Reader:
    def __init__(self, b: bytes):
        self.b, self.o = b, 0

    def align(self, n):
        self.o += (-self.o) % n

    def prim(self, t):
        fmt, size = _PRIM[t]
        self.align(size)
        v = struct.unpack_from(fmt, self.b, self.o)[0]
        self.o += size
        return v

    def value(self, t):
        if isinstance(t, tuple):
            if t[0] == "seq":
                n = self.prim("uint32")
                return [self.value(t[1]) for _ in range(n)]
            return [self.value(t[1]) for _ in range(t[2])]
        if t == "string":
            n = self.prim("uint32")
            s = self.b[self.o:self.o + n - 1].decode("utf-8")
            self.o += n
            return s
        if t in _PRIM:
            return self.prim(t)
        m = Msg(t)
        for name, ft in SCHEMAS[t]:
            m.fields[name] = self.value(ft)
        return m


CDR_LE_HEADER = b"\x00\x01\x00\x00"


def encode(msg: Msg) -> bytes:
    w = _Writer()
    w.value(msg.type, msg)
    return CDR_LE_HEADER + bytes(w.b)


def decode(type_name: str, data: bytes) -> Msg:
    if data[:4] != CDR_LE_HEADER:
        raise ValueError("not a little-endian CDR stream")
    return _Reader(data[4:]).value(type_name)


# ----------------------------------------------------------------------------- how the node fills them
def _header(stamp: float, frame_id: str = "") -> Msg:
    h = default("std_msgs/Header")
    sec = int(np.floor(stamp))
    h.fields["stamp"].fields.update(sec=sec, nanosec=int(round((stamp - sec) * 1e9)) % 1_000_000_000)
    h.fields["frame_id"] = frame_id
    return h


def fill_particle(particles_6p, weights, stamp: float) -> Msg:
    """publish_particle_info (OdometryPipeline.cpp:940-962): rows of the [6·P] vector, then the weights."""
    v = np.asarray(particles_6p, float).reshape(-1)
    P = v.size // 6
    m = default("stein_msgs/SteinParticle")
    m.fields["header"] = _header(stamp)
    for i, name in enumerate(("x", "y", "z", "roll", "pitch", "yaw")):
        m.fields[name] = v[i * P:(i + 1) * P].tolist()
    m.fields["weights"] = np.asarray(weights, float).reshape(-1).tolist()
    return m


def fill_runtime(steinicp_time: float, preprocessing_time: float, stamp: float, knn_time: float = 0.0, update_time: float = 0.0,
                 finish_iter: float = 0.0) -> Msg:
    m = default("stein_msgs/Runtime")
    m.fields["header"] = _header(stamp)
    m.fields.update(steinicp_time=float(steinicp_time), preprocessing_time=float(preprocessing_time), knn_time=float(knn_time),
                    update_time=float(update_time), finish_iter=float(finish_iter))
    return m


def fill_variance(var_icp, stamp: float, var_mean_filtered=None, var_maxsliding_filtered=None, var_random_walk=None) -> Msg:
    m = default("stein_msgs/Variance")
    m.fields["header"] = _header(stamp)
    z = [0.0] * 6
    m.fields.update(var_icp=[float(x) for x in np.asarray(var_icp).reshape(6)],
                    var_mean_filtered=[float(x) for x in (z if var_mean_filtered is None else var_mean_filtered)],
                    var_maxsliding_filtered=[float(x) for x in (z if var_maxsliding_filtered is None else var_maxsliding_filtered)],
                    var_random_walk=[float(x) for x in (z if var_random_walk is None else var_random_walk)])
    return m


def msg_definition(type_name: str) -> str:
    """Text of a ROS 2 ``.msg`` definition generated from the schema."""
    lines = []
    for name, t in SCHEMAS[type_name]:
        if isinstance(t, tuple):
            base = t[1].split("/")[-1] if t[1].startswith("stein_msgs/") else t[1]
            ts = f"{base}[]" if t[0] == "seq" else f"{base}[{t[2]}]"
        else:
            ts = t
        lines.append(f"{ts} {name}")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    import argparse
    import os
    ap = argparse.ArgumentParser()
    ap.add_argument("--emit", metavar="DIR", required=True, help="write stein_msgs/msg/*.msg generated from the schemas")
    a = ap.parse_args()
    os.makedirs(os.path.join(a.emit, "msg"), exist_ok=True)
    for tn in SCHEMAS:
        if tn.startswith("stein_msgs/"):
            with open(os.path.join(a.emit, "msg", tn.split("/")[1] + ".msg"), "w") as f:
                f.write(msg_definition(tn))
