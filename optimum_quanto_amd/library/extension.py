"""Native-extension registry for the ``quanto::`` operator library.

Mirrors the public surface of the reference registry (``register_extension``, ``get_extension``,
``is_extension_available`` - optimum/quanto/library/extensions/extension.py:58-86) but the extension
behind it is a torch-free C-ABI shared library (``include/quanto_hip.h``) bound with ctypes, built
in-tree with hipcc for gfx950 instead of JIT-compiled through ``torch.utils.cpp_extension``.
"""
import ctypes
import os
import shutil
import subprocess
import threading
from typing import Dict, List, Optional

__all__ = ["NativeLibrary", "register_extension", "get_extension", "is_extension_available"]


class NativeLibrary:
    """A lazily loaded (and, when a compiler is present, lazily built) C-ABI shared library.

    Same life cycle as the reference ``Extension`` (extension.py:12-55): nothing happens at import time,
    the first access to ``.lib`` loads - or builds then loads - the binary.
    """

    def __init__(self, name: str, root_dir: str, lib_path: str, sources: List[str], make_dir: Optional[str] = None):
        self.name = name
        self.root_dir = root_dir
        self.lib_path = lib_path
        self.sources = [os.path.join(root_dir, s) for s in sources]
        self.make_dir = make_dir or root_dir
        self._cdll = None
        self._lock = threading.Lock()

    # -- build ------------------------------------------------------------------------------------
    def is_stale(self) -> bool:
        if not os.path.exists(self.lib_path):
            return True
        built = os.path.getmtime(self.lib_path)
        return any(os.path.exists(s) and os.path.getmtime(s) > built for s in self.sources)

    def build(self, force: bool = False) -> str:
        """Compile the library for gfx950 with hipcc (cross-compiles without a GPU)."""
        if not force and not self.is_stale():
            return self.lib_path
        if shutil.which("make") is None or not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
            raise RuntimeError(f"{self.name}: cannot build {self.lib_path}: hipcc/make not found")
        cmd = ["make", "-C", self.make_dir, f"-j{os.cpu_count() or 4}"]
        if force:
            cmd.insert(1, "-B")
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"{self.name}: build failed\n{proc.stdout[-4000:]}\n{proc.stderr[-4000:]}")
        return self.lib_path

    # -- load -------------------------------------------------------------------------------------
    @property
    def available(self) -> bool:
        return os.path.exists(self.lib_path)

    @property
    def cdll(self) -> ctypes.CDLL:
        if self._cdll is None:
            with self._lock:
                if self._cdll is None:
                    if not os.path.exists(self.lib_path):
                        # same behaviour as the reference JIT: build on first use when a toolchain exists
                        self.build()
                    self._cdll = ctypes.CDLL(self.lib_path)
        return self._cdll


_extensions: Dict[str, object] = {}


def register_extension(extension) -> None:
    if extension.name in _extensions:
        raise ValueError(f"extension {extension.name} is already registered")
    _extensions[extension.name] = extension


def get_extension(extension_type: str):
    """Return a registered extension (KeyError if unknown)."""
    return _extensions[extension_type]


def is_extension_available(extension_type: str) -> bool:
    """True when an extension of that name was registered for this platform."""
    return extension_type in _extensions
