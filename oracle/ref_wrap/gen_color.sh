#!/bin/sh
# oracle/ref_wrap/gen_color.sh REFSRC GENDIR "EXTRACT" -- TEST INFRASTRUCTURE ONLY.
# Lifts the colour-module pixel functions out of the reference sources into the scratch dir.
set -e
R="$1"; G="$2"; X="$3"
$X $R/colorprofiles/conversion.h $G/conversion_h.inc DT_CONVERSION_LUT_SAMPLES dt_colorspaces_conversion_hook_t
$X $R/colorprofiles/conversion.c $G/conversion.inc dt_colorspaces_conversion_t _clamp_unit _apply_target_curves _apply_matrix
$X $R/colorprofiles/iop_profile.h $G/iop_profile.inc extrapolate_lut eval_exp dt_ioppr_eval_trc
$X $R/iop/colorin.c $G/colorin.inc apply_blue_mapping
$X $R/iop/channelmixerrgb.c $G/channelmixerrgb.inc INVERSE_SQRT_3 dt_iop_channelmixer_rgb_version_t gamut_mapping luma_chroma loop_switch
