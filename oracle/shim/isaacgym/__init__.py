# Stand-in package for the absent, closed-source Isaac Gym distribution.
# Only ``torch_utils`` is provided; see that module's header.
