#!/bin/sh
# set_targets.sh — defines the classification targets for one database directory, with the calling convention and the
# files of CuCLARK's script of the same name (set_targets.sh:32-126):
#   ./set_targets.sh <database directory> <bacteria|viruses|human|custom>+ [--species|--genus|--family|--order|--class|--phylum]
# For every selected database it gathers the metadata (make_metadata.sh), appends "<file>\t<taxid at the rank>" to
# <dir>/targets.txt (exe/getTargetsDef), creates <dir>/<db1>_..._<rank>_canonical/ for the k-mer database, and writes
# ./.settings ("-T <dir>/targets.txt" and "-D <dir>/<subdir>/") for classify_metagenome.sh, ./.DBDirectory for the
# housekeeping scripts and <dir>/files_excluded.txt for the sequences without a taxonomy ID.
HERE=$(dirname "$0")
if [ $# -lt 2 ]; then
  echo "Usage: $0 <Directory path> <Databases: bacteria, viruses, human or custom>+ <taxonomy rank: --phylum, --class, --order, --family, --genus or --species (default)>"
  exit 0
fi
DBDR=$1
if [ -z "$DBDR" ]; then
  echo "The database directory must not be empty."
  exit 1
fi
if [ ! -d "$DBDR" ]; then
  echo "Selected directory not found. The program will create it."
  mkdir -m 775 "$DBDR"
  if [ ! -d "$DBDR" ]; then
    echo "Failed to create the directory (please check the name of directory $DBDR and whether it exists). The program will abort."
    exit 1
  fi
fi
echo "$DBDR" > .DBDirectory

RANK=0
shift
for arg in "$@"; do
  case "$arg" in
    --species) RANK=0; break ;;
    --genus) RANK=1; break ;;
    --family) RANK=2; break ;;
    --order) RANK=3; break ;;
    --class) RANK=4; break ;;
    --phylum) RANK=5; break ;;
    --*) echo "Failed to recognize this parameter: $arg"; exit 1 ;;
  esac
done

for stale in "$DBDR/targets.txt" "$DBDR/.tmp" "$DBDR/files_excluded.txt" .settings files_excluded.txt; do
  [ -f "$stale" ] && unlink "$stale"
done
touch "$DBDR/targets.txt"
SUBDB=""
for db in "$@"; do
  case "$db" in --*) continue ;; esac
  printf "Collecting metadata of %s... " "$db"
  "$HERE/make_metadata.sh" "$db" "$DBDR"
  [ -s "$DBDR/.$db" ] || exit 1
  [ -f "$DBDR/.taxondata" ] || exit 1
  echo "done."
  if [ -s "$DBDR/.$db.fileToTaxIDs" ]; then
    "$HERE/exe/getTargetsDef" "$DBDR/.$db.fileToTaxIDs" $RANK >> "$DBDR/targets.txt"
    SUBDB="${SUBDB}${db}_"
    cat files_excluded.txt >> "$DBDR/.tmp"
    unlink files_excluded.txt
  fi
done
SUBDB="${SUBDB}${RANK}_canonical"
echo "-T $DBDR/targets.txt" > .settings
if [ ! -d "$DBDR/$SUBDB" ]; then
  echo "Creating directory to store discriminative k-mers: $DBDR/$SUBDB"
  mkdir -m 775 "$DBDR/$SUBDB"
fi
echo "-D $DBDR/$SUBDB/" >> .settings
if [ -s "$DBDR/.tmp" ]; then
  mv "$DBDR/.tmp" "$DBDR/files_excluded.txt"
elif [ -f "$DBDR/.tmp" ]; then
  unlink "$DBDR/.tmp"
fi
