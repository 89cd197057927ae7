"""python -m wayne_amd -p params.yml  ==  python -m wayne_amd.run_visit -p params.yml  (the reference's `wayne`)."""
from .run_visit import run

if __name__ == "__main__":
    run()
