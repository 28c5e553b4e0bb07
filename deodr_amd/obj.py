"""``read_obj`` / ``save_obj`` with the signatures of ``deodr.obj`` (deodr/obj.py): the triangle-mesh subset of Wavefront OBJ that
DEODR's examples and fitters start from -- ``v`` and ``f`` records (``f`` corners may carry ``/vt/vn`` indices, which are
ignored here; relative, i.e. negative, indices count back from the vertices read so far).  Host-side convenience, no device
involved."""

import numpy as np


def read_obj(filename):
    """-> (faces [T,3] int, vertices [V,3] float64), like the reference's loader"""
    vertices, faces = [], []
    with open(filename) as stream:
        pending = ""
        for raw in stream:
            line = pending + raw.rstrip("\n")
            if line.endswith("\\"):  # continuation
                pending = line[:-1] + " "
                continue
            pending = ""
            fields = line.split()
            if not fields:
                continue
            if fields[0] == "v":
                vertices.append([float(x) for x in fields[1:4]])
            elif fields[0] == "f":
                corners = [int(corner.split("/")[0]) for corner in fields[1:]]
                if len(corners) != 3:  # (the reference returns an [n, 4] array that its TriMesh then rejects; say so here)
                    raise ValueError(f"{filename}: face with {len(corners)} corners -- only triangle meshes are supported")
                faces.append([c - 1 if c > 0 else len(vertices) + c for c in corners])
    return np.array(faces, dtype=np.int64).reshape(-1, 3), np.array(vertices, dtype=np.float64).reshape(-1, 3)


def save_obj(filename, vertices, faces):
    with open(filename, "w") as stream:
        for x, y, z in np.asarray(vertices, dtype=np.float64):
            stream.write(f"v {x:.17g} {y:.17g} {z:.17g}\n")
        for a, b, c in np.asarray(faces, dtype=np.int64):
            stream.write(f"f {a + 1} {b + 1} {c + 1}\n")
