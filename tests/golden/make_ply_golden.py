"""Generates tests/golden/ply_attributes.json: the vertex attribute list the REFERENCE's PLY writer produces
(gs3dgs/scene/gaussian_model.py:220-232, GaussianModel.construct_list_of_attributes) for three model shapes.

Runs in the build container only (the reference tree is not on the GPU box).  The reference's gaussian_model module
cannot be imported here (its package pulls plyfile and simple_knn), so this script loads just that one method with `ast`
from the file where it lies, executes it on a stand-in object with the right tensor shapes, and stores the OUTPUT.
    python tests/golden/make_ply_golden.py
"""
import ast
import json
import os
import types

REF = "/root/reference/gs3dgs/scene/gaussian_model.py"
tree = ast.parse(open(REF).read())
fn = next(n for cls in tree.body if isinstance(cls, ast.ClassDef) and cls.name == "GaussianModel"
          for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "construct_list_of_attributes")
ns = {}
exec(compile(ast.Module([fn], []), REF, "exec"), ns)


class Shape:
    def __init__(self, *s):
        self.shape = s


out = {}
for name, (k, s) in {"sh3_3d": (16, 3), "sh0_3d": (1, 3), "sh3_2d": (16, 2)}.items():
    me = types.SimpleNamespace(_features_dc=Shape(5, 1, 3), _features_rest=Shape(5, k - 1, 3), _scaling=Shape(5, s), _rotation=Shape(5, 4))
    out[name] = {"sh_coeffs": k, "scale_dims": s, "attributes": ns["construct_list_of_attributes"](me)}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ply_attributes.json"), "w"), indent=0)
print({k: len(v["attributes"]) for k, v in out.items()})
