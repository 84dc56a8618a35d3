"""Ahead-of-time build of on-demand step-kernel specialisations (include/mpcqp.h: mpcqp_prebuild).

    python -m mpcqp.prebuild [manifest]        (default: spec_manifest.txt next to this file)

Needs hipcc, not a GPU.  The objects land in the specialisation cache (MPCQP_CACHE_DIR, default lib/spec_cache);
mpcqp_prepare on the target machine loads them and verifies each once against the runtime-dimension kernel."""
import os
import sys

from . import api

DEFAULT_MANIFEST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spec_manifest.txt")


def read_manifest(path=None):
    """[(nu, ny, nxhat, Hp, Hc, neps, row_groups)] of a manifest file (# comments, blank lines ignored)."""
    shapes = []
    with open(path or DEFAULT_MANIFEST) as f:
        for n, line in enumerate(f, 1):
            line = line.split("#", 1)[0].split()
            if not line:
                continue
            if len(line) != 7:
                raise ValueError(f"{path or DEFAULT_MANIFEST}:{n}: expected `nu ny nxhat Hp Hc neps row_groups(hex)`")
            shapes.append(tuple(int(v) for v in line[:6]) + (int(line[6], 16),))
    return shapes


def prebuild(nu, ny, nxhat, Hp, Hc, neps=1, row_groups=0, lib=None):
    """Compile (if not cached) the specialisation of one shape.  Returns the MPCQP_KERNEL_* kind steps of that shape
    will run on (AOT: the shape is compiled into the library, nothing to build; GENERIC: the shape has no
    specialisation -- custom constraints, nZ~ > 128); raises MpcqpError with the build log path on failure."""
    lib = lib or api.load_library()
    d = api.Dims(batch=1, nxhat=nxhat, nu=nu, ny=ny, nd=0, Hp=Hp, Hc=Hc, neps=neps)
    rc = lib.mpcqp_prebuild(d, row_groups)
    if rc < 0:
        why = lib.mpcqp_last_build_error()
        raise api.MpcqpError(f"mpcqp_prebuild({nu},{ny},{nxhat},{Hp},{Hc},{neps},0x{row_groups:x}): "
                             f"{lib.mpcqp_strerror(rc).decode()}: {(why or b'').decode()}")
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    shapes = read_manifest(argv[0] if argv else None)
    for s in shapes:
        kind = prebuild(*s[:6], row_groups=s[6])
        print("nu=%d ny=%d nxhat=%d Hp=%d Hc=%d neps=%d rows=0x%x" % s, "->", {0: "runtime-dimension kernel (no specialisation for this shape)",
              1: "compiled into the library", 2: "specialisation in the cache", 3: "small-problem kernel"}.get(kind, kind))
    return 0


if __name__ == "__main__":
    sys.exit(main())
