"""Shim: the reference's mortal/config.py imports `toml`, which this image lacks; tomli provides the reader."""
import tomli


def load(f):
    if hasattr(f, "read"):
        data = f.read()
        return tomli.loads(data if isinstance(data, str) else data.decode())
    with open(f, "rb") as fh:
        return tomli.load(fh)


def loads(s):
    return tomli.loads(s)
