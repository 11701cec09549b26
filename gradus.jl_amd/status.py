"""StatusCodes -- src/Gradus.jl:59-64 (EnumX, declaration order)."""
import enum


class StatusCodes(enum.IntEnum):
    OutOfDomain = 0
    WithinInnerBoundary = 1
    IntersectedWithGeometry = 2
    NoStatus = 3
