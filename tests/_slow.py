"""The slow tier of the GPU suite (`-m "gpu and slow"`).

`pytest -m gpu -x -q` is what the driver runs at round end under a fixed limit; round 4's suite took 537-553 s of it on a fresh box.  The
cases below are the long tail of that run (seconds measured on MI355X, round 5, cold on-disk code-object cache: gpurun_out/r05_gpu_full.log
-> profiles/r05_gpu_test_durations.txt): every one of them repeats, at a bigger size or a longer iteration count, a check the fast tier keeps
for the same kernel (same test function, smaller parameters) - no SURVEY 8 row and no BASELINE configuration loses its fast-tier coverage.
They are deselected unless the `-m` expression names `slow` (or SPCIES_RUN_SLOW=1); the builder runs them through gpurun and commits the log
(profiles/r05_gpu_slow_tier.log)."""

SLOW = {
    # the bit-exact STREAM kernels at the largest plants (the CPU oracle's run time, mostly): (7, 3) and (3, 5) stay in the fast tier
    "test_stream_is_bit_exact_for_any_plant_size[36-4-5-laxMPC-eadmm]": 23.0,
    "test_stream_is_bit_exact_for_any_plant_size[36-4-5-laxMPC-admm]": 12.6,
    "test_stream_is_bit_exact_for_any_plant_size[36-4-5-laxMPC-fista]": 9.6,
    "test_stream_is_bit_exact_for_any_plant_size[29-6-6-equMPC-admm]": 7.0,
    "test_stream_is_bit_exact_for_any_plant_size[29-6-6-equMPC-fista]": 6.8,
    "test_stream_is_bit_exact_for_any_plant_size[29-6-6-equMPC-eadmm]": 3.0,
    # converging runs (thousands of iterations) of the one-lane-per-instance STREAM variants at the big shapes; the 200-iteration runs of
    # the same (config, variant) stay
    "test_mpct_cs_seeded_batch_vs_oracle[stream-C2_cs-24-overrides3]": 22.3,
    "test_hmpc_seeded_batch_vs_oracle[stream-C5_HMPC_SADMM-12-overrides5]": 21.4,
    "test_mpct_general_qr_seeded_batch_vs_oracle[C4_nd-40-overrides2-stream]": 18.0,
    "test_mpct_seeded_batch_vs_oracle[stream-C4-70-overrides2]": 15.3,
    "test_hip_time_varying_fista_vs_oracle[C2_equ_FISTA-50-overrides3-stream]": 9.1,
    "test_mpct_general_qr_seeded_batch_vs_oracle[C4_nd-90-overrides1-stream]": 7.4,
    "test_hmpc_coupled_split_vs_oracle[stream-C1_HMPCcc_soc-33]": 6.8,
    "test_hmpc_coupled_split_vs_oracle[stream-C1_HMPCcc-40]": 6.5,
    "test_mpct_cs_seeded_batch_vs_oracle[stream-C1_MPCT_cs_vec-40-overrides1]": 5.6,
    "test_vector_rho_and_var_bounds_past_the_block_programs[C2_lax_N30_gen-24-overrides3]": 10.0,
    # the n = 20 forms of admm_r (17 s of hiprtc each on a cold cache); the n = 12, N = 30 forms of the same tests stay
    "test_vector_rho_and_var_bounds_past_the_block_programs[C4_lax_ADMM_gen-48-overrides2]": 20.2,
    "test_admm_past_the_register_file_vs_oracle[C4_lax_ADMM-40-overrides4]": 18.8,
    "test_admm_r_plain_and_unit_box_coordinates[C2_equ_N30-40-overrides1]": 13.8,
    "test_admm_past_the_register_file_vs_oracle[C2_equ_N30-40-overrides1]": 8.8,
    # arbitrary-shape sweeps: the largest shape of each (a fresh hiprtc specialisation of 5-10 s); the other shapes stay
    "test_time_varying_any_plant_size[13-3-5-laxMPC-ADMM]": 12.6,
    "test_time_varying_any_plant_size[10-4-7-laxMPC-FISTA]": 6.1,
    "test_mfma4r_plants_with_many_inputs_or_up_to_32_rows[26-3-5-admm]": 8.0,
    "test_admm_r_arbitrary_shapes[18-3-9-laxMPC]": 7.9,
    "test_mfma4r_plants_with_many_inputs_or_up_to_32_rows[26-3-5-fista]": 5.7,
    # later in round 5 (fast tier 423 s with the any-plant-size STREAM cases added): duplicates of shapes the fast tier keeps
    "test_vector_rho_and_var_bounds_past_the_block_programs[C2_equ_N30_gen-48-overrides1]": 9.0,   # lax at the same shape stays
    "test_admm_r_plain_and_unit_box_coordinates[C2_lax_N30-48-overrides0]": 7.9,                  # C1_lax stays; C2_lax_N30 runs in test_admm_past_the_register_file
    "test_time_varying_any_plant_size[10-4-7-equMPC-ADMM]": 8.7,                                  # (9, 2) lax and (8, 3) equ FISTA stay
    "test_time_varying_plants_past_the_register_file[12-6-5-equMPC-ADMM]": 8.8,                   # (20, 4) lax ADMM, (17, 3) equ FISTA, (20, 2) lax FISTA stay
}


def is_slow(nodeid):
    return nodeid.split("::")[-1] in SLOW
