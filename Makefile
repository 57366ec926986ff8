# Convenience targets; __graft_entry__.build() runs the same three builds.
.PHONY: all lib host oracle test clean
all: lib host oracle
lib:
	$(MAKE) -C linreg-mpc_amd/csrc
host: lib
	$(MAKE) -C linreg-mpc_amd/host
oracle:
	$(MAKE) -C oracle
test: all
	python -m pytest tests -q -m "not gpu"
clean:
	$(MAKE) -C linreg-mpc_amd/csrc clean
	$(MAKE) -C linreg-mpc_amd/host clean
	$(MAKE) -C oracle clean
