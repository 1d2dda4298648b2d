from reverb_amd.ctc_align import adjust_model_time_offset, ctc_align  # noqa: F401
