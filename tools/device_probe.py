"""Device speed probes for the tools/ scripts (tools/micro/device_probe.hip, built on first use; not part of the product library)."""
import ctypes, os, subprocess
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        src = os.path.join(_HERE, "micro", "device_probe.hip"); so = os.path.join(_HERE, "micro", "libdeviceprobe.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, src], check=True)
        import torch  # noqa: F401  (one HIP runtime per process: bind to the one torch mapped)
        _LIB = ctypes.CDLL(so)
    return _LIB


def shader_clock_mhz(device: int = 0) -> float:
    mhz = ctypes.c_double(0.0)
    if _lib().probe_clock_mhz(int(device), ctypes.byref(mhz)):
        raise RuntimeError("probe_clock_mhz failed")
    return mhz.value


def speed_probe(device: int = 0):
    """(iterations per microsecond of a dependent integer multiply-add chain in one wave, GB/s of a 256 MiB streaming copy)."""
    alu, gbs = ctypes.c_double(0.0), ctypes.c_double(0.0)
    if _lib().probe_speed(int(device), ctypes.byref(alu), ctypes.byref(gbs)):
        raise RuntimeError("probe_speed failed")
    return alu.value, gbs.value
