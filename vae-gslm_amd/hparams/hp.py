"""YAML-backed nested hyper-parameter namespace.

API-compatible with the reference's ``hparams.hp.Hparams`` (hparams/hp.py:9-66):
attribute access, ``get`` / ``has`` / ``check_arg_in_hparams`` (raises
``ValueError`` on a missing key) / ``merge`` / ``save`` and the ``from_*``
constructors.  Unknown top-level blocks (e.g. this build's optional ``hip:``)
are carried along untouched.
"""
from __future__ import annotations

import json
from argparse import Namespace
from types import SimpleNamespace
from typing import Any, Mapping

import yaml


def _wrap(obj):
    """Recursively turn dicts into Hparams (lists are walked, scalars kept)."""
    if isinstance(obj, dict):
        return Hparams(**{k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [_wrap(v) for v in obj]
    return obj


def _unwrap(obj):
    if isinstance(obj, Hparams):
        return {k: _unwrap(v) for k, v in vars(obj).items()}
    if isinstance(obj, (list, tuple)):
        return [_unwrap(v) for v in obj]
    return obj


class Hparams(SimpleNamespace):
    def __init__(self, *args, **kwargs):
        super().__init__(**kwargs)

    # ---- presence / lookup
    def check_arg_in_hparams(self, *names: str) -> None:
        for n in names:
            if n not in vars(self):
                raise ValueError(f"{n} not specifed in the hyperapramer: {self}")

    def has(self, name: str) -> bool:
        return name in vars(self)

    def get(self, name: str, default=None):
        return vars(self).get(name, default)

    def merge(self, other: "Hparams") -> "Hparams":
        return Hparams(**vars(self), **vars(other))

    def __eq__(self, other) -> bool:
        return vars(self) == vars(other)

    def __repr__(self) -> str:
        return repr(vars(self))

    # ---- (de)serialisation
    def to_dict(self) -> Mapping[str, Any]:
        return _unwrap(self)

    def save(self, path: str) -> None:
        with open(path, "w") as f:
            yaml.dump(self.to_dict(), f)

    @classmethod
    def from_dict(cls, d: Mapping[str, Any]) -> "Hparams":
        return _wrap(dict(d))

    @classmethod
    def from_yamlfile(cls, yamlfile: str) -> "Hparams":
        with open(yamlfile, "r") as f:
            return _wrap(yaml.safe_load(f))

    @classmethod
    def from_jsonfile(cls, jsonfile: str) -> "Hparams":
        with open(jsonfile, "r") as f:
            return _wrap(json.load(f))

    @classmethod
    def from_json(cls, json_s: str) -> "Hparams":
        return _wrap(json.loads(json_s))

    @classmethod
    def from_argparse(cls, args: Namespace) -> "Hparams":
        return _wrap(dict(vars(args)))
