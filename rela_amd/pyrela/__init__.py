"""Python side of the Ape-X / R2D2 training entry points (learner, agents, synthetic envs)."""
