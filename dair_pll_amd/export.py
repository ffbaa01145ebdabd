"""Reporting side of the drop-in: learned parameters back to URDF / OBJ / scalar summaries.

Host-only and off the hot path (the reference does this with Drake's parser state and torch on the CPU:
``dair_pll/urdf_utils.py:255-384``, ``dair_pll/deep_support_function.py:19-123``,
``dair_pll/multibody_terms.py:536-582``).  Nothing here launches a kernel.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np

DRAKE_URL = 'https://drake.mit.edu/'
_PROX = '{' + DRAKE_URL + '}proximity_properties'
_MU = '{' + DRAKE_URL + '}mu_static'
INERTIA_ATTRIBUTES = ('ixx', 'iyy', 'izz', 'ixy', 'ixz', 'iyz')
MESH_FILE = 'test.obj'  # the name the reference writes (urdf_utils.py:248)


# ---- convex mesh of a support function ------------------------------------------------------------
def surface_directions() -> np.ndarray:
    """Unit directions through the boundary nodes of an 8 x 8 x 8 grid on [-1, 1]^3
    (``deep_support_function.py:12-15``): 296 directions, deterministic order (x slowest)."""
    line = np.linspace(-1.0, 1.0, 8)
    grid = np.stack(np.meshgrid(line, line, line, indexing='ij'), -1).reshape(-1, 3)
    surface = grid[np.abs(grid).max(axis=1) >= 1.0]
    return surface / np.linalg.norm(surface, axis=1, keepdims=True)


def outward_normals(vertices: np.ndarray, faces: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Unit normals pointing away from the vertex centroid, whether each face had to be flipped, and the
    plane offsets ``n . v`` (``deep_support_function.py:56-90``)."""
    v_a, v_b, v_c = (vertices[faces[:, i]] for i in range(3))
    normals = np.cross(v_b - v_a, v_c - v_a)
    normals = normals / np.linalg.norm(normals, axis=1, keepdims=True)
    backwards = ((v_a - vertices.mean(axis=0)) * normals).sum(axis=1) < 0.0
    normals[backwards] *= -1.0
    return normals, backwards, (v_a * normals).sum(axis=1)


def extract_mesh(support_function: Callable[[np.ndarray], np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """Vertices = distinct support points over :func:`surface_directions`, faces = their convex hull with
    counter-clockwise winding seen from outside (``deep_support_function.py:93-123``)."""
    from scipy.spatial import ConvexHull
    points = np.asarray(support_function(surface_directions()), dtype=np.float64)
    seen, unique = set(), []
    for point in points:
        key = point.tobytes()
        if key not in seen:
            seen.add(key)
            unique.append(point)
    vertices = np.stack(unique)
    faces = np.asarray(ConvexHull(vertices).simplices, dtype=np.int64)
    _, backwards, _ = outward_normals(vertices, faces)
    faces[backwards] = faces[backwards][:, ::-1]
    return vertices, faces


def mesh_to_obj(vertices: np.ndarray, faces: np.ndarray) -> str:
    """Wavefront text with one normal per face, ``f a//n b//n c//n`` (``deep_support_function.py:19-52``)."""
    normals, _, _ = outward_normals(vertices, faces)
    lines = ['v ' + ' '.join(repr(float(c)) for c in vertex) for vertex in vertices] + ['', '']
    lines += ['vn ' + ' '.join(repr(float(c)) for c in normal) for normal in normals] + ['', '']
    lines += ['f ' + ' '.join(f'{int(index) + 1}//{face_index + 1}' for index in face) for face_index, face in enumerate(faces)]
    return '\n'.join(lines) + '\n'


# ---- URDF ------------------------------------------------------------------------------------------
def _find_or_add(parent: ET.Element, tag: str, defaults: Dict[str, str]) -> ET.Element:
    """The reference fills missing elements with zeroed defaults (``urdf_utils.py:125-183``)."""
    child = parent.find(tag)
    if child is None:
        child = ET.SubElement(parent, tag, dict(defaults))
    return child


def _set_shape(geometry_element: ET.Element, tag: str, attributes: Dict[str, str]) -> None:
    shape = geometry_element.find(tag)
    if shape is None:
        shape = ET.SubElement(geometry_element, tag)
    shape.attrib = dict(attributes)


def fill_link(link: ET.Element, pi_cm: Sequence[float], shapes: List[Tuple[Tuple[str, Dict[str, str]], float]]) -> None:
    """Writes one body's learned values into its ``<link>`` (``urdf_utils.py:255-314``): mass, centre of
    mass (``pi_cm[1:4] / m``), central inertia, and for every ``(shape, mu)`` of ``shapes`` the link's next
    ``<collision>`` shape with its ``drake:mu_static``; the first shape is also the ``<visual>`` one."""
    zero3 = {'xyz': '0. 0. 0.', 'rpy': '0. 0. 0.'}
    inertial = _find_or_add(link, 'inertial', {})
    _find_or_add(inertial, 'mass', {'value': '0.'}).set('value', repr(float(pi_cm[0])))
    origin = _find_or_add(inertial, 'origin', zero3)
    origin.set('xyz', ' '.join(repr(float(c) / float(pi_cm[0])) for c in pi_cm[1:4]))
    # the written tensor is the body-frame one, so the inertial frame is the body's: the reference leaves a source
    # URDF's inertial rpy in place (urdf_utils.py:290 sets xyz only), which would turn the tensor it wrote once more
    origin.set('rpy', zero3['rpy'])
    _find_or_add(inertial, 'inertia', {}).attrib = {k: repr(float(v)) for k, v in zip(INERTIA_ATTRIBUTES, pi_cm[4:])}
    collisions = link.findall('collision')
    while len(collisions) < len(shapes):
        collisions.append(ET.SubElement(link, 'collision', {}))
    for position, ((shape, mu), collision) in enumerate(zip(shapes, collisions)):
        holders = (collision, _find_or_add(link, 'visual', {})) if position == 0 else (collision,)
        for holder in holders:
            _set_shape(_find_or_add(holder, 'geometry', {}), shape[0], shape[1])
        properties = _find_or_add(collision, _PROX, {})
        _find_or_add(properties, _MU, {'value': '0.'}).set('value', repr(float(mu)))


def render_urdf(source_path: str, bodies: List[Tuple[str, Sequence[float], List[Tuple[Tuple[str, Dict[str, str]], float]]]]) -> str:
    """The source URDF with every listed ``(link name, pi_cm, [(shape, mu), ...])`` written into it, as text
    (``urdf_utils.py:317-384``); links not listed (no inertia: the world) are left alone."""
    tree = ET.parse(source_path)
    by_name = {name: (pi_cm, shapes) for name, pi_cm, shapes in bodies}
    for element in tree.iter():
        if element.tag == 'link' and element.get('name') in by_name:
            pi_cm, shapes = by_name[element.get('name')]
            fill_link(element, pi_cm, shapes)
    ET.register_namespace('drake', DRAKE_URL)
    return '<?xml version="1.0"?>\n' + ET.tostring(tree.getroot(), encoding='utf-8').decode('utf-8')


def save_string(path: str, text: str) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w', encoding='utf8') as handle:
        handle.write(text)
