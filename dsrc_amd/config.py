"""Compression settings of one archive, as the host side hands them to the C ABI (include/dsrc_gpu.h: dsrcgpu_settings,
dsrcgpu_dataset).  Orders, not command-line levels; `from_levels` is the reference's mapping."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class Config:
    dna_order: int = 0
    quality_order: int = 0
    lossy: bool = False
    crc: bool = False
    quality_offset: int = 33
    plus_repetition: bool = False
    color_space: bool = False
    tag_flags: int = 0

    @staticmethod
    def from_levels(d: int, q: int, lossy: bool = False, crc: bool = False, offset: int = 33) -> "Config":
        # IDsrcOperator::GetCompressionSettings (reference src/DsrcOperator.h:74-90)
        return Config(dna_order=3 * d, quality_order=(3 * q if lossy else q), lossy=lossy, crc=crc, quality_offset=offset)
