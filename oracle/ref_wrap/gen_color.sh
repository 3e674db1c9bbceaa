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
$X $R/colorprofiles/iop_profile.h $G/iop_profile_info.inc dt_iop_order_iccprofile_info_t _apply_trc dt_ioppr_get_rgb_matrix_luminance
$X $R/colorprofiles/iop_profile.c $G/iop_profile_c.inc _apply_tonecurves _transform_rgb_to_lab_matrix _transform_lab_to_rgb_matrix
$X $R/iop/filmicrgb.c $G/filmicrgb.inc INVERSE_SQRT_3 SAFETY_MARGIN CIE_Y_1931_to_CIE_Y_2006 ORDER_4 ORDER_3 \
  dt_iop_filmicrgb_methods_type_t dt_iop_filmicrgb_curve_type_t dt_iop_filmicrgb_colorscience_type_t \
  dt_iop_filmicrgb_spline_version_type_t dt_iop_filmic_noise_distribution_t _filmic_is_agx \
  dt_iop_filmic_rgb_spline_t dt_iop_filmicrgb_params_t dt_iop_filmicrgb_data_t \
  dt_iop_filmicrgb_v3_geometry_t dt_iop_filmicrgb_v3_nodes_t filmic_v3_compute_geometry filmic_v3_compute_nodes_from_legacy \
  pixel_rgb_norm_power_simd get_pixel_norm_simd get_pixel_norm filmic_desaturate_v1 filmic_desaturate_v2 linear_saturation \
  filmic_split_v1 filmic_split_v2_v3 filmic_chroma_v1 filmic_chroma_v2_v3 log_tonemapping exp_tonemapping_v2 filmic_spline \
  pipe_RGB_to_Ych_simd Ych_to_pipe_RGB_simd filmic_desaturate_v4 clip_chroma_white_raw clip_chroma_white \
  clip_chroma_black clip_chroma gamut_check_Yrg_filmic_simd gamut_check_RGB_simd gamut_mapping_simd \
  filmic_v4_prepare_matrices dt_iop_filmicrgb_simd_matrices_t filmic_prepare_simd_matrices \
  norm_tone_mapping_v4_simd RGB_tone_mapping_v4_simd filmic_chroma_v4 filmic_split_v4 filmic_v5 \
  _filmic_agx_xyz_D50_to_Yrg _filmic_agx_Yrg_to_xyz_D50 _mat3_identity _filmic_agx_build_displaced \
  filmic_agx_prepare_bracket filmic_agx_compress_negatives filmic_agx filmic_sigmoid_scale \
  dt_iop_filmic_rgb_compute_spline
$X $R/colorprofiles/iop_profile.h $G/iop_profile_xyz.inc dt_ioppr_rgb_matrix_to_xyz
$X $R/develop/blend.h $G/blend_h.inc dt_develop_blend_colorspace_t dt_develop_blend_mode_t dt_develop_mask_mode_t \
  dt_develop_mask_combine_mode_t dt_develop_mask_feathering_guide_t dt_develop_blendif_channels_t dt_develop_blend_params_t \
  DEVELOP_BLENDIF_PARAMETER_ITEMS
$X $R/develop/blend.c $G/blend_c.inc dt_develop_blendif_process_parameters dt_develop_blendif_init_masking_profile \
  _develop_blend_process_mask_tone_curve _develop_mask_post_processing _develop_mask_get_post_operations \
  _develop_blend_process_feather _detail_mask_threshold
$X $R/develop/masks/detail.c $G/masks_detail.inc dt_masks_extend_border dt_masks_blur_9x9_coeff FAST_BLUR_9 dt_masks_blur_9x9 \
  dt_masks_calc_rawdetail_mask calcBlendFactor dt_masks_calc_detail_mask
$X $R/develop/blends/blendif_rgb_jzczhz.c $G/blendif_rgb_jzczhz.inc DT_BLENDIF_RGB_CH DT_BLENDIF_RGB_BCH \
  _blendif_compute_factor _blendif_gray _blendif_rgb_red _blendif_rgb_green _blendif_rgb_blue _blendif_jzczhz \
  _blendif_combine_channels dt_develop_blendif_rgb_jzczhz_make_mask _blend_normal _blend_multiply _blend_add _blend_subtract \
  _blend_subtract_inverse _blend_difference _blend_divide _blend_divide_inverse _blend_average _blend_geometric_mean \
  _blend_harmonic_mean _blend_chromaticity _blend_luminance _blend_RGB_R _blend_RGB_G _blend_RGB_B _choose_blend_func _copy_mask \
  dt_develop_blendif_rgb_jzczhz_blend
$X $R/colorprofiles/iop_profile.h $G/iop_profile_lab.inc dt_ioppr_rgb_matrix_to_lab
$X $R/develop/blends/blendif_lab.c $G/blendif_lab.inc DT_BLENDIF_LAB_CH DT_BLENDIF_LAB_BCH _CLAMP _CLAMP_XYZ \
  _blendif_compute_factor _blendif_lab_l _blendif_lab_a _blendif_lab_b _blendif_lch _blendif_combine_channels \
  dt_develop_blendif_lab_make_mask _blend_Lab_scale _blend_Lab_rescale _blend_normal_bounded _blend_normal_unbounded \
  _blend_lighten _blend_darken _blend_multiply _blend_average _blend_add _blend_subtract _blend_difference \
  _blend_difference2 _blend_screen _blend_overlay _blend_softlight _blend_hardlight _blend_vividlight \
  _blend_linearlight _blend_pinlight _blend_lightness _blend_chromaticity _blend_hue _blend_color _blend_coloradjust \
  _blend_Lab_lightness _blend_Lab_a _blend_Lab_b _blend_Lab_color _choose_blend_func _copy_mask \
  dt_develop_blendif_lab_blend
$X $R/develop/blends/blendif_raw.c $G/blendif_raw.inc dt_develop_blendif_raw_make_mask _blend_normal_bounded \
  _blend_normal_unbounded _blend_lighten _blend_darken _blend_multiply _blend_average _blend_add _blend_subtract \
  _blend_difference _blend_screen _blend_overlay _blend_softlight _blend_hardlight _blend_vividlight _blend_linearlight \
  _blend_pinlight _choose_blend_func dt_develop_blendif_raw_blend
$X $R/develop/blends/blendif_rgb_hsl.c $G/blendif_rgb_hsl.inc DT_BLENDIF_RGB_CH DT_BLENDIF_RGB_BCH _CLAMP_XYZ _PX_COPY \
  _blendif_compute_factor _blendif_gray _blendif_gray_fb _blendif_rgb_red _blendif_rgb_green _blendif_rgb_blue _blendif_hsl \
  _blendif_combine_channels dt_develop_blendif_rgb_hsl_make_mask _blend_normal_bounded _blend_normal_unbounded \
  _blend_lighten _blend_darken _blend_multiply _blend_average _blend_add _blend_subtract _blend_difference _blend_screen \
  _blend_overlay _blend_softlight _blend_hardlight _blend_vividlight _blend_linearlight _blend_pinlight _blend_lightness \
  _blend_chromaticity _blend_hue _blend_color _blend_coloradjust _blend_HSV_value _blend_HSV_color _blend_RGB_R \
  _blend_RGB_G _blend_RGB_B _choose_blend_func _copy_mask dt_develop_blendif_rgb_hsl_blend
