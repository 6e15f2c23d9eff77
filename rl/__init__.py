"""Import-path compatibility package: checkpoints written by torch.save(module) in osudrl/apex reference the globals
`rl.policies.actor.Gaussian_FF_Actor` and `rl.policies.critic.FF_V`.  These modules are this
repository's own implementation of those classes (same attribute names and forward signatures) so that actor.pt /
critic.pt written here load in the reference's tools and vice versa."""
