"""Wavefront OBJ -> GroupShape tree, mirroring lib/src/obj_parser.rs (host-side input format of the mesh path).

Kept faithful to the reference, quirks included, because they decide what the renderer is fed:
  * every line counts as "ignored" (obj_parser.rs:203: the counter is bumped for all lines);
  * all `v` records must precede the first `f`; at the first face (or at end of input) the vertices are
    normalised into the [-1, 1] cube around their bounding box centre (:250-263), in f32;
  * faces are fan-triangulated (:265-291); when the first face vertex carries a normal index the triangles
    are SmoothTriangles -- whose normals are looked up with the VERTEX index, not the normal index (:281-283);
  * `g name` starts a new group; faces before any `g` go to the group named "" (:167-176).
`api` is the module providing GroupShape / Triangle / SmoothTriangle (this package by default; tests also pass
the oracle's mirror so both sides build their own trees from the same text).
"""
import numpy as np

f32 = np.float32


class ParseError(ValueError):
    """obj_parser.rs:56-66; `kind` names the enum variant."""

    def __init__(self, kind, message):
        super().__init__(message)
        self.kind = kind


class ObjParseResults:
    def __init__(self, num_ignored_lines, vertices, normals, groups, api):
        self.num_ignored_lines = num_ignored_lines
        self.vertices, self.normals = vertices, normals
        self.groups = groups  # insertion-ordered dict name -> GroupShape (None once taken)
        self._api = api

    def get_default_group(self):
        return None if self.groups is None else self.groups.get("")

    def get_group(self, name):
        return None if self.groups is None else self.groups.get(name)

    def take_all_as_group(self):
        """obj_parser.rs:33-53.  (The reference drains a HashMap, i.e. in unspecified order; here: file order.)"""
        if self.groups is None:
            return None
        groups, self.groups = self.groups, None
        if len(groups) == 1:
            return next(iter(groups.values()))
        all_as_group = self._api.GroupShape()
        for g in groups.values():
            all_as_group.add_child(g)
        return all_as_group


def _parse_f32(tok):
    try:
        return f32(tok)  # f32::from_str: correctly rounded decimal -> f32, as numpy's parse-to-double-then-round is
    except ValueError:                       # for every literal short enough to be exact in f64 (all OBJ files here)
        raise ParseError("ParseFloatError", "invalid float literal")


def _parse_face(face_string):
    """obj_parser.rs:223-247 -> (vertex, texture, normal)"""
    elements = []
    for x in face_string.split("/"):
        if x == "":
            elements.append(None)
        else:
            if not x.isdigit():
                raise ParseError("ParseIntError", "invalid digit found in string")
            elements.append(int(x))
    if elements[0] is None:
        raise ParseError("MalformedFace", "Missing vertex index")
    return elements[0], (elements[1] if len(elements) > 1 else None), (elements[2] if len(elements) > 2 else None)


def _normalize_vertices(vertices):
    """obj_parser.rs:250-263, all in f32"""
    if len(vertices) < 2:
        return
    pts = np.array(vertices[1:], dtype=f32)
    mn, mx = np.fmin.reduce(pts[:, :3], axis=0), np.fmax.reduce(pts[:, :3], axis=0)
    span = (mx - mn).astype(f32)
    scale = np.fmax(span[0], np.fmax(span[1], span[2])) / f32(2.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        for v in vertices[1:]:
            for k in range(3):
                v[k] = (v[k] - (mn[k] + span[k] / f32(2.0))) / scale


def _fan_triangulation(api, vertices, normals, face_specs):
    """obj_parser.rs:265-291"""
    triangles = []
    smooth = face_specs[0][2] is not None
    for index in range(1, len(face_specs) - 1):
        v1, v2, v3 = (vertices[face_specs[k][0]] for k in (0, index, index + 1))
        if smooth:
            n1, n2, n3 = (normals[face_specs[k][0]] for k in (0, index, index + 1))  # sic: indexed by .vertex
            triangles.append(api.SmoothTriangle(v1.copy(), v2.copy(), v3.copy(), n1.copy(), n2.copy(), n3.copy()))
        else:
            triangles.append(api.Triangle(v1.copy(), v2.copy(), v3.copy()))
    return triangles


def parse_obj(text, api=None):
    """parse_obj (obj_parser.rs:100-215).  `text`: str, bytes or an iterable of lines."""
    if api is None:
        from . import api as api_module
        api = api_module
    if isinstance(text, bytes):
        text = text.decode()
    lines = text.split("\n") if isinstance(text, str) else list(text)
    if isinstance(text, str) and lines and lines[-1] == "":
        lines.pop()  # BufRead::lines does not yield an empty line after a trailing newline
    num_ignored_lines = 0
    vertices = [np.array([0, 0, 0, 1], dtype=f32)]
    normals = [np.array([0, 0, 0, 1], dtype=f32)]
    groups = {}
    current_group = None
    normalization_finished = False
    for index, line in enumerate(lines):
        elements = line.strip().split()
        head = elements[0] if elements else None
        if head == "v":
            if normalization_finished:
                raise ParseError("UnexpectedSymbol", "Found vertex at line %d; vertices must all be specified before any "
                                 "faces are specified (so that they may be normalized before any faces are created)" % index)
            coordinates = [_parse_f32(x) for x in elements[1:]]
            if len(coordinates) != 3:
                raise ParseError("MalformedVertex", "Wrong number of coordinates in vertex at line %d; expected 3, found %d"
                                 % (index, len(coordinates)))
            vertices.append(np.array(coordinates + [f32(1.0)], dtype=f32))
        elif head == "vn":
            coordinates = [_parse_f32(x) for x in elements[1:]]
            if len(coordinates) != 3:
                raise ParseError("MalformedNormal", "Wrong number of coordinates in normal vector at line %d; expected 3, "
                                 "found %d" % (index, len(coordinates)))
            normals.append(np.array(coordinates + [f32(0.0)], dtype=f32))
        elif head == "f":
            if not normalization_finished:
                _normalize_vertices(vertices)
                normalization_finished = True
            face_specs = [_parse_face(x) for x in elements[1:]]
            if len(face_specs) < 3:
                raise ParseError("MalformedFace", "Not enough vertices to form a face at line %d; expected 3, found %d"
                                 % (index, len(face_specs)))
            if current_group is None:
                groups[""] = api.GroupShape()
                current_group = groups[""]
            for triangle in _fan_triangulation(api, vertices, normals, face_specs):
                current_group.add_child(triangle)
        elif head == "g":
            if len(elements) < 2:
                raise ParseError("MalformedGroupDeclaration", "Missing group name on line %d" % index)
            groups[elements[1]] = api.GroupShape()
            current_group = groups[elements[1]]
        num_ignored_lines += 1
    if not normalization_finished:
        _normalize_vertices(vertices)
    return ObjParseResults(num_ignored_lines, vertices, normals, groups, api)
