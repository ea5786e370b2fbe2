from .map import GeometricMap, Map  # noqa: F401
