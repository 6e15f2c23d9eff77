// placeholder until the physics kernel lands (next commit)
#include "apx_common.h"
extern "C" void apx_env_default_cfg(apx_env_cfg* c) { if (c) { *c = apx_env_cfg{}; c->n_envs = 4096; c->simrate = 50; c->dynamics_randomization = 1; c->have_incentive = 1; c->max_traj_len = 400; c->pgs_iters = 50; } }
#define NI(name, ...) extern "C" int name(__VA_ARGS__) { apx_set_error(#name ": not implemented yet"); return APX_E_STATE; }
NI(apx_env_create, const apx_env_cfg*, apx_env_t**)
NI(apx_env_destroy, apx_env_t*)
NI(apx_env_reset, apx_env_t*, const uint8_t*, float*, void*)
NI(apx_env_step, apx_env_t*, const float*, float*, float*, uint8_t*, float*, int, void*)
NI(apx_env_get_state, apx_env_t*, float*, float*, void*)
NI(apx_env_set_state, apx_env_t*, const float*, const float*, void*)
NI(apx_env_get_field, apx_env_t*, const char*, float*, void*)
NI(apx_env_set_field, apx_env_t*, const char*, const float*, void*)
