"""Layered settings dictionaries (mirror of ``magmap.settings.profiles.SettingsDict``).

A profile is a dict of defaults plus named *modifiers*; ``add_profiles("a,b,file.yaml")``
applies modifiers left to right, later ones winning, nested dicts merged rather than
replaced (reference magmap/settings/profiles.py:122-244).  YAML files are looked up under
``profiles/`` first, then as given (:185-204), and can be hot-reloaded when their mtime
changes (:246-270).  ``is_identical_settings`` compares a key subset across profiles
(:272-297).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import yaml


class SettingsDict(dict):
    PATH_PROFILES = "profiles"
    NAME_KEY = "settings_name"
    DEFAULT_NAME = "default"
    _YAML_EXTS = (".yml", ".yaml")

    def __init__(self, *args, **kwargs):
        super().__init__()
        self[self.NAME_KEY] = self.DEFAULT_NAME
        #: named modifiers: ``{name: {key: value}}``
        self.profiles: Dict[str, Dict] = {}
        #: YAML path -> mtime when loaded
        self.timestamps: Dict[str, float] = {}
        self.delimiter = ","
        self.update(*args, **kwargs)

    # -- applying modifiers ------------------------------------------------------------
    def modify_settings(self, mods: Dict) -> None:
        for key, val in mods.items():
            if key in self:
                cur = self[key]
                if isinstance(cur, dict) and isinstance(val, dict):
                    cur.update(val)
                else:
                    self[key] = val
            elif hasattr(self, key):
                cur = getattr(self, key)
                if isinstance(cur, dict) and isinstance(val, dict):
                    cur.update(val)
                else:
                    setattr(self, key, val)
            # unknown keys are ignored, as the reference does

    def _yaml_path(self, name: str) -> Optional[str]:
        cand = os.path.join(self.PATH_PROFILES, name)
        if os.path.exists(cand):
            return cand
        return name if os.path.exists(name) else None

    def get_profile(self, name: str) -> Optional[Dict]:
        if os.path.splitext(name)[1].lower() in self._YAML_EXTS:
            path = self._yaml_path(name)
            if path is None:
                print(name, "profile file not found, skipped")
                return None
            self.timestamps[path] = os.path.getmtime(path)
            mods: Dict = {}
            with open(path) as f:
                for doc in yaml.safe_load_all(f):
                    if doc:
                        mods.update(doc)
            return mods
        if name == self.DEFAULT_NAME:
            return type(self)()
        if name not in self.profiles:
            print(name, "profile not found, skipped")
            return None
        return self.profiles[name]

    def add_profiles(self, names_str: str) -> None:
        for name in names_str.split(self.delimiter):
            mods = self.get_profile(name)
            if mods:
                self[self.NAME_KEY] += self.delimiter + name
                self.modify_settings(mods)

    # -- hot reload --------------------------------------------------------------------
    def check_file_changed(self) -> bool:
        return any(ts < os.path.getmtime(path) for path, ts in self.timestamps.items())

    def refresh_profile(self, check_timestamp: bool = False) -> None:
        if check_timestamp and not self.check_file_changed():
            return
        names = self[self.NAME_KEY]
        self.__init__()
        self.add_profiles(names)

    @staticmethod
    def is_identical_settings(profs: Iterable[dict], keys: Iterable[str]) -> bool:
        profs = list(profs)
        keys = list(keys)
        same = all(profs[0][k] == p[k] for p in profs[1:] for k in keys) if profs else True
        print("Block settings are identical" if same else "Block settings are not identical")
        return same
