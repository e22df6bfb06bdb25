/* Test-only entry points of libsslam_frontend.so -- NOT part of the drop-in boundary (include/sslam_frontend.h is).  tests/ call them; a product
 * caller has no reason to. */
#ifndef SSLAM_TESTING_H
#define SSLAM_TESTING_H
#ifdef __cplusplus
extern "C" {
#endif

/* Multi-GPU group code (sslam_group_create / _create_rank, reference: the batch-of-frames mode of BASELINE configs[4]) on a box with ONE GPU:
 * while `on` is nonzero, NEW groups bind an in-process stand-in for the RCCL entry points (host-mediated device-to-device copies with NCCL's matching rules;
 * same process only) and may hold more members than GPUs are visible -- so that the group code's multi-member paths (a host thread per member, uneven tails,
 * the collective error agreement, grouped send / receive to the root) execute with real device buffers.  It says nothing about xGMI.  Process-wide; returns the
 * previous setting.  (Round 4 selected the stand-in with the environment variable SSLAM_GROUP_FAKE_RCCL; an explicit call cannot be set by accident.) */
int sslam_testing_use_rccl_standin(int on);

#ifdef __cplusplus
}
#endif
#endif
