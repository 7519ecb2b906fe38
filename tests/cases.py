"""Named rtl_fm configurations used by fixtures and parity tests.

Each case is (name, cfg-overrides, signal kwargs).  CLI equivalents follow
SURVEY.md §8: C1 `-M fm -s 240k -m 2.2M -A fast`, C2 `-M fm -s 150k -m 1.3M -F 0`,
C3 `-M fm -s 16k -F 9 -E deemp`, WBFM `-M wbfm`.
"""
from rtlsdr_amd.capi import (ATAN_FAST, ATAN_LUT, ATAN_STD, MODE_AM, MODE_FM, MODE_LSB,
                             MODE_RAW, MODE_USB, RESAMPLE_ARBITRARY, RESAMPLE_LOW_PASS_REAL,
                             RtlfmCfg)

WB = dict(fs=2.4e6, dev_hz=75e3)
NB = dict(fs=1.024e6, dev_hz=2.5e3)

CASES = [
    # name, cfg overrides, signal
    ("c1_boxcar10_fast", dict(downsample=10, custom_atan=ATAN_FAST, rate_out=240000), WB),
    ("c2_p4_std", dict(downsample=16, downsample_passes=4, rate_out=150000), WB),
    ("c2_p4_fir9_std", dict(downsample=16, downsample_passes=4, comp_fir_size=9, rate_out=150000), WB),
    ("c2_p4_lut", dict(downsample=16, downsample_passes=4, custom_atan=ATAN_LUT, rate_out=150000), WB),
    ("c2_p4_fast_a40", dict(downsample=16, downsample_passes=4, custom_atan=ATAN_FAST, rate_out=150000),
     dict(fs=2.4e6, dev_hz=75e3, amplitude=40.0)),
    ("c3_p6_fir9_deemph", dict(downsample=64, downsample_passes=6, comp_fir_size=9, deemph=1,
                               deemph_a=2, rate_out=16000), NB),
    ("c3_p6_fir9_deemph_up22050", dict(downsample=64, downsample_passes=6, comp_fir_size=9, deemph=1,
                                      deemph_a=2, rate_out=16000, rate_out2=22050,
                                      resampler=RESAMPLE_ARBITRARY), NB),
    ("c3_p6_arb_down8000", dict(downsample=64, downsample_passes=6, rate_out=16000, rate_out2=8000,
                                resampler=RESAMPLE_ARBITRARY), NB),
    ("wbfm_preset", dict(downsample=6, custom_atan=ATAN_FAST, deemph=1, deemph_a=13, rate_out=170000,
                         rate_out2=32000, resampler=RESAMPLE_LOW_PASS_REAL),
     dict(fs=1.02e6, dev_hz=75e3)),
    ("p1_std", dict(downsample=2, downsample_passes=1), NB),
    ("p2_fir9", dict(downsample=4, downsample_passes=2, comp_fir_size=9), NB),
    ("p3_lut", dict(downsample=8, downsample_passes=3, custom_atan=ATAN_LUT), NB),
    ("p5_std_dc", dict(downsample=32, downsample_passes=5, dc_block_audio=1), NB),
    ("p7_fir9", dict(downsample=128, downsample_passes=7, comp_fir_size=9), NB),
    ("p8_std", dict(downsample=256, downsample_passes=8), NB),
    ("p10_fir9", dict(downsample=1024, downsample_passes=10, comp_fir_size=9), NB),
    ("p4_rdc", dict(downsample=16, downsample_passes=4, dc_block_raw=1), WB),
    ("p4_offset_tuning", dict(downsample=16, downsample_passes=4, offset_tuning=1), WB),
    ("p4_post4", dict(downsample=16, downsample_passes=4, post_downsample=4), WB),
    ("p4_squelch", dict(downsample=16, downsample_passes=4, squelch_level=2000), WB),
    ("p4_squelch_open", dict(downsample=16, downsample_passes=4, squelch_level=100), WB),
    ("am_p4", dict(mode=MODE_AM, downsample=16, downsample_passes=4, output_scale=16), WB),
    ("usb_box8", dict(mode=MODE_USB, downsample=8, output_scale=32), WB),
    ("lsb_p3", dict(mode=MODE_LSB, downsample=8, downsample_passes=3, output_scale=32), WB),
    ("raw_p2", dict(mode=MODE_RAW, downsample=4, downsample_passes=2), WB),
    ("raw_box1", dict(mode=MODE_RAW, downsample=1), WB),
    ("box7_std_lpr", dict(downsample=7, rate_out=48000, rate_out2=11025,
                          resampler=RESAMPLE_LOW_PASS_REAL), WB),
    ("box256_std", dict(downsample=256), WB),
    # per-buffer stages behind a boxcar that does not divide the buffer (the reference's own default
    # plans: 8192 % 42, % 84, % 6, % 334 != 0), SURVEY.md §8 a14-a18
    ("box42_dc", dict(downsample=42, rate_out=24000, dc_block_audio=1), dict(fs=1.008e6, dev_hz=2.5e3)),  # rtl_fm -s 24k -E dc
    ("box84_am_dc", dict(mode=MODE_AM, downsample=84, rate_out=12000, output_scale=3, dc_block_audio=1),
     dict(fs=1.008e6, dev_hz=2.5e3)),  # rtl_fm -M am -s 12k -E dc
    ("box6_wbfm_dc", dict(downsample=6, custom_atan=ATAN_FAST, deemph=1, deemph_a=13, rate_out=170000,
                          rate_out2=32000, resampler=RESAMPLE_LOW_PASS_REAL, dc_block_audio=1),
     dict(fs=1.02e6, dev_hz=75e3)),  # rtl_fm -M wbfm -E dc
    ("box334_usb", dict(mode=MODE_USB, downsample=334, rate_out=3000, output_scale=1), dict(fs=1.002e6, dev_hz=1e3)),  # -M usb -s 3k
    ("box42_dc_arb_up32000", dict(downsample=42, rate_out=24000, dc_block_audio=1, rate_out2=32000,
                                  resampler=RESAMPLE_ARBITRARY), dict(fs=1.008e6, dev_hz=2.5e3)),
    ("box10_deemph_arb_down96000", dict(downsample=10, custom_atan=ATAN_FAST, deemph=1, deemph_a=19, rate_out=240000,
                                        rate_out2=96000, resampler=RESAMPLE_ARBITRARY), WB),
    ("box1000_std_squelch", dict(downsample=1000, rate_out=1000, squelch_level=50), dict(fs=1.0e6, dev_hz=200.0)),
    # the everyday scanner line, rtl_fm -M fm -s 12k -l 50: the default boxcar (/84) + the power squelch, on a keyed
    # carrier (loud, silent and half-and-half buffers); and -M raw behind the boxcar
    ("box84_fm_squelch50", dict(downsample=84, rate_out=12000, squelch_level=50),
     dict(fs=1.008e6, dev_hz=2.5e3, amplitude=0.8, quiet=(40000, 22000))),
    ("raw_box10", dict(mode=MODE_RAW, downsample=10, rate_out=240000), WB),
    ("box10_rdc_fast", dict(downsample=10, custom_atan=ATAN_FAST, dc_block_raw=1, rate_out=240000), WB),  # rtl_fm -s 240k -A fast -E rdc
]


def make_cfg(overrides: dict, block_len: int, max_blocks: int = 1) -> RtlfmCfg:
    return RtlfmCfg.default(block_len=block_len, max_blocks=max_blocks, **overrides)


def case(name: str):
    for n, o, s in CASES:
        if n == name:
            return o, s
    raise KeyError(name)
